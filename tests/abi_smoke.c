/* The C-ABI from C: include/kccot.h must compile as plain C99 (it is what a maintainer's cgo / JNI / N-API / ctypes stub
 * binds), every declared entry point must be addressable with its declared type, and the entry points that need no GPU must
 * behave as documented when called through dlopen.  Built and run by tests/test_abi.py (CPU tier). */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "kccot.h"

#define CHECK(cond, msg) do { if (!(cond)) { fprintf(stderr, "abi_smoke: %s\n", msg); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: abi_smoke <path to libkccot.so>\n"); return 2; }
    void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "abi_smoke: dlopen failed: %s\n", dlerror()); return 1; }
    /* typed through the header's own declarations: a signature drift between header and call site is a compile error */
    __typeof__(kccot_version)* version = (__typeof__(kccot_version)*)dlsym(h, "kccot_version");
    __typeof__(kccot_last_error)* last_error = (__typeof__(kccot_last_error)*)dlsym(h, "kccot_last_error");
    __typeof__(kccot_set_option)* set_option = (__typeof__(kccot_set_option)*)dlsym(h, "kccot_set_option");
    __typeof__(kccot_get_option)* get_option = (__typeof__(kccot_get_option)*)dlsym(h, "kccot_get_option");
    __typeof__(kccot_option_count)* option_count = (__typeof__(kccot_option_count)*)dlsym(h, "kccot_option_count");
    __typeof__(kccot_option_name)* option_name = (__typeof__(kccot_option_name)*)dlsym(h, "kccot_option_name");
    __typeof__(kccot_pairwise_cost3_workspace_bytes)* ws_bytes =
        (__typeof__(kccot_pairwise_cost3_workspace_bytes)*)dlsym(h, "kccot_pairwise_cost3_workspace_bytes");
    __typeof__(kccot_pairwise_cost3_f32)* cost3 = (__typeof__(kccot_pairwise_cost3_f32)*)dlsym(h, "kccot_pairwise_cost3_f32");
    CHECK(version && last_error && set_option && get_option && option_count && option_name && ws_bytes && cost3, "missing symbol");
    CHECK(version() == KCCOT_VERSION, "library and header disagree on KCCOT_VERSION");
    int n = option_count(), v = -1, i;
    CHECK(n >= 10, "option table too small");
    for (i = 0; i < n; ++i) {
        const char* name = option_name(i);
        CHECK(name && name[0], "unnamed option");
        CHECK(get_option(name, &v) == 0, "get_option failed on a listed option");
        CHECK(set_option(name, v) == 0, "set_option rejected an option's own value");
    }
    CHECK(option_name(n) == NULL && option_name(-1) == NULL, "option_name out of range must be NULL");
    CHECK(set_option("no_such_option", 1) == KCCOT_EINVAL, "unknown option must be KCCOT_EINVAL");
    CHECK(strstr(last_error(), "no_such_option") != NULL, "kccot_last_error must name the offending option");
    CHECK(set_option("sinkhorn_shortcut", 7) == KCCOT_EINVAL, "out-of-range value must be KCCOT_EINVAL");
    CHECK(ws_bytes(64, 122880) > 0 && ws_bytes(0, 122880) == 0, "workspace query");
    /* argument checks come before any device work: a null pointer is rejected without a GPU */
    CHECK(cost3(NULL, NULL, 64, 122880, 1.0f / 15.0f, NULL, NULL, NULL, NULL, 30, 8, 0u, NULL, NULL, 0, NULL) == KCCOT_EINVAL,
          "null pointers must be KCCOT_EINVAL");
    printf("abi_smoke ok: version %d, %d options\n", version(), n);
    dlclose(h);
    return 0;
}
