"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol that
include/kccot.h declares, rejects bad arguments before any launch, and the Python host mirror keeps
the reference's signatures.  No compute call is made (no GPU here)."""
import ctypes
import inspect
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "kccot.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kccot_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from kccotgan_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(_lib.lib, s), "libkccot.so does not export %s" % s
    assert sorted(_lib.SIGNATURES) == syms, "ctypes table and header disagree"
    assert _lib.lib.kccot_version() == 301


def test_argument_validation_happens_before_any_launch():
    from kccotgan_amd import _lib
    lib = _lib.lib
    one = ctypes.c_void_p(16)   # never dereferenced: every call below is rejected on its arguments
    assert lib.kccot_pairwise_cost_f32(None, one, 4, 4, 8, 1.0, None, None, None, None, 1, 1, 0, one, None, 0, None) == _lib.EINVAL
    assert b"null" in lib.kccot_last_error()
    assert lib.kccot_pairwise_cost_f32(one, one, 0, 4, 8, 1.0, None, None, None, None, 1, 1, 0, one, None, 0, None) == _lib.EINVAL
    assert lib.kccot_pairwise_cost_f32(one, one, 4, 4, 8, 1.0, one, None, None, None, 2, 2, 0, one, None, 0, None) == _lib.EINVAL
    # workspace too small is reported, not overrun
    need = lib.kccot_pairwise_cost_workspace_bytes(4, 4, 64)
    assert need > 0
    assert lib.kccot_pairwise_cost_f32(one, one, 4, 4, 64, 1.0, None, None, None, None, 1, 1, 0, one, one, 16, None) == _lib.EWORKSPACE
    assert lib.kccot_sinkhorn_fwd_f32(one, 1, 0, 1.0, 10, 10, 0.01, 0, None, None, one, one, None, None, 0, None) == _lib.EINVAL
    assert lib.kccot_sinkhorn_fwd_f32(one, 1, 8, -1.0, 10, 10, 0.01, 0, None, None, one, one, None, None, 0, None) == _lib.EINVAL
    assert lib.kccot_sinkhorn_fwd_f32(one, 1, 4096, 1.0, 10, 10, 0.01, 0, None, None, one, one, None, None, 0, None) == _lib.EUNSUPPORTED
    assert lib.kccot_sinkhorn_fwd_f32(one, 1, 512, 1.0, 10, 10, 0.01, 0, None, None, one, one, None, None, 0, None) == _lib.EWORKSPACE
    assert lib.kccot_sinkhorn_workspace_bytes(3, 64) == 0 and lib.kccot_sinkhorn_workspace_bytes(3, 512) >= 2 * 3 * 512 * 512 * 4
    assert lib.kccot_smooth_fwd_f32(one, 2, 8, 3, 8, 1, 5.0, 3, _lib.SMOOTH_T, one, one, one, 1 << 30, None) == _lib.EINVAL  # radius >= T
    assert lib.kccot_martingale_fwd_f32(one, 0, 4, 4, 1.0, 1.0, one, None) == _lib.EINVAL
    # round 2 entry points: fused solve + sweep, scaled cost backward, solver status
    assert lib.kccot_sinkhorn_fused_eligible(64, 100) == 1 and lib.kccot_sinkhorn_fused_eligible(8, 100) == 1
    assert lib.kccot_sinkhorn_fused_eligible(256, 100) == 0 and lib.kccot_sinkhorn_fused_eligible(64, 5000) == 0
    assert lib.kccot_sinkhorn_divergence_fused_f32(one, 64, 1.0, 100, 100, 0.01, one, one, one, one, None, None) == _lib.EINVAL
    assert lib.kccot_sinkhorn_divergence_fused_f32(one, 256, 1.0, 100, 100, 0.01, one, one, one, one, one, None) == _lib.EUNSUPPORTED
    assert lib.kccot_sinkhorn_divergence_fused_f32(one, 64, 0.0, 100, 100, 0.01, one, one, one, one, one, None) == _lib.EINVAL
    assert lib.kccot_pairwise_cost3_bwd_scaled_f32(one, None, one, one, 8, 64, 1.0, None, None, None, None, 1, 1, one, None, None,
                                                   None, None, one, 1 << 20, None) == _lib.EINVAL
    assert lib.kccot_sinkhorn_status(None, 3, None) == _lib.EINVAL


def test_workspace_queries_are_monotone_and_cover_both_cost_paths():
    from kccotgan_amd import _lib
    lib = _lib.lib
    a = lib.kccot_pairwise_cost3_workspace_bytes(64, 122880)
    b = lib.kccot_pairwise_cost3_workspace_bytes(64, 2 * 122880)
    assert 0 < a <= b
    assert lib.kccot_pairwise_cost3_workspace_bytes(0, 10) == 0
    assert lib.kccot_smooth_workspace_bytes(2, 8, 4, 8, 1) >= 2 * 8 * 4 * 8 * 4


def test_host_mirror_keeps_reference_signatures():
    """Names, argument order and defaults of gan_utils.py:6,21,46,75,124,168,179,204 and
    data_utils.py:479,503,523,552,584."""
    from kccotgan_amd import gan_utils as g, data_utils as d

    def params(f):
        return [(p.name, p.default if p.default is not inspect._empty else None)
                for p in inspect.signature(f).parameters.values() if p.kind != p.KEYWORD_ONLY]

    assert params(g.cost_xy) == [("x", None), ("y", None), ("scaling_coef", None)]
    assert params(g.modified_cost) == [("x", None), ("y", None), ("h", None), ("M", None), ("scaling_coef", None)]
    assert [n for n, _ in params(g.bi_causal_modified_cost)] == ["x", "y", "hy", "Mx", "hx", "My", "scaling_coef"]
    assert params(g.benchmark_sinkhorn) == [("x", None), ("y", None), ("scaling_coef", None), ("epsilon", 1.0),
                                            ("L", 10), ("Lmin", 10)]
    assert params(g.compute_sinkhorn) == [("x", None), ("y", None), ("hy", None), ("Mx", None),
                                          ("scaling_coef", None), ("hx", None), ("My", None), ("epsilon", 1.0),
                                          ("L", 100), ("bi_causal", False)]
    assert [n for n, _ in params(g.compute_sinkhorn_loss)] == [
        "f_real", "f_fake", "scaling_coef", "sinkhorn_eps", "sinkhorn_l", "h_fake", "m_real", "h_real",
        "m_fake", "video"]
    assert params(g.scale_invariante_martingale_regularization) == [("M", None), ("reg_lam", None),
                                                                    ("scaling_coef", None)]
    assert params(g.compute_N) == [("M", None)]
    ks = d.KernelSmoothing()
    assert (ks.temporal_radius, ks.spatial_radius) == (3, 4)            # data_utils.py:479-481 defaults 6, 8
    ks = d.KernelSmoothing(temporal_kernel_size=6, spatial_kernel_size=6)   # kernel_train.py:216
    assert (ks.temporal_radius, ks.spatial_radius) == (3, 3)
    assert params(ks.annealing_sigma) == [("init_sigma", None), ("step", None), ("decay_steps", 500),
                                          ("decay_rate", 0.975)]
    assert abs(ks.annealing_sigma(5.0, 1000) - 5.0 * 0.975 ** 2) < 1e-12
    for m in ("temporal_convolution", "spatial_convolution", "gaussian_convolution3D"):
        assert [n for n, _ in params(getattr(ks, m))] == ["inputs", "sigma"]


def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU tensor -- never route through the oracle."""
    import torch
    from kccotgan_amd import gan_utils as g, _lib
    x = torch.zeros(2, 3, 4)
    with pytest.raises(_lib.KccotError):
        g.cost_xy(x, x, 1.0)
    src = ""
    for root, _, files in os.walk(os.path.join(ROOT, "kccotgan_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src += open(os.path.join(root, f)).read()
    assert "import oracle" not in src and "from oracle" not in src


def test_option_table_is_the_only_run_time_switchboard():
    """kccot_set_option / kccot_get_option: every option the header documents exists with the documented default, unknown
    names and out-of-range values are rejected, and no entry point reads the environment (the sources contain one getenv:
    KCCOT_OPTIONS, read once at first use; the fault-injection hook's is compiled into the diagnostic twin only)."""
    from kccotgan_amd import _lib
    text = open(os.path.join(ROOT, "include", "kccot.h")).read()
    documented = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"^ \*   ([a-z0-9_]+)\s+(-?\d+)\s{2,}\S", text, flags=re.M))
    names = _lib.option_names()
    assert sorted(documented) == sorted(names), (sorted(documented), sorted(names))
    for k in names:
        assert _lib.get_option(k) == documented[k], k           # the CPU tier runs on defaults
    assert _lib.lib.kccot_set_option(b"no_such_option", 1) == _lib.EINVAL and b"unknown" in _lib.lib.kccot_last_error()
    assert _lib.lib.kccot_set_option(b"gram_f32", 2) == _lib.EINVAL and b"outside" in _lib.lib.kccot_last_error()
    assert _lib.lib.kccot_set_option(None, 1) == _lib.EINVAL
    assert _lib.lib.kccot_option_name(-1) is None and _lib.lib.kccot_option_name(len(names)) is None
    with _lib.options(sinkhorn_shortcut=0, sinkhorn_fused_max_n=128):
        assert _lib.get_option("sinkhorn_shortcut") == 0 and _lib.get_option("sinkhorn_fused_max_n") == 128
        assert _lib.lib.kccot_sinkhorn_fused_eligible(128, 20) == 1
        with _lib.options(sinkhorn_fused=0):
            assert _lib.lib.kccot_sinkhorn_fused_eligible(64, 100) == 0
    assert _lib.get_option("sinkhorn_shortcut") == 1 and _lib.lib.kccot_sinkhorn_fused_eligible(128, 20) == 0
    src = os.path.join(ROOT, "kccotgan_amd", "csrc")
    hits = []
    for fn in sorted(os.listdir(src)):
        if fn.endswith((".hip", ".h")):
            body = open(os.path.join(src, fn)).read()
            body = re.sub(r"#ifdef KCCOT_DIAG.*?#e(?:lse|ndif)", "", body, flags=re.S)
            hits += [(fn, m.group(0)) for m in re.finditer(r'getenv\("[A-Z_]+"\)', body)]
    assert hits == [("api.hip", 'getenv("KCCOT_OPTIONS")')], hits


def test_header_is_plain_c_and_the_abi_works_from_c(tmp_path):
    """include/kccot.h compiles as strict C99 (what a cgo / JNI / N-API stub includes), and a C program that dlopens the
    library -- typed through the header's own declarations -- sees the documented behaviour of every entry point that
    needs no GPU: version, the option table, error codes and the error string, workspace queries, argument checks."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    probe = tmp_path / "hdr.c"
    probe.write_text('#include "kccot.h"\nint main(void) { return 0; }\n')
    inc = os.path.join(ROOT, "include")
    r = subprocess.run([gcc, "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", inc, "-fsyntax-only", str(probe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    exe = tmp_path / "abi_smoke"
    r = subprocess.run([gcc, "-std=gnu99", "-Wall", "-Wextra", "-Werror", "-I", inc, "-o", str(exe),
                        os.path.join(ROOT, "tests", "abi_smoke.c"), "-ldl"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe), os.path.join(ROOT, "kccotgan_amd", "csrc", "libkccot.so")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi_smoke ok" in r.stdout


# ---- ISA pins of the fence-free hand-offs --------------------------------------------------------
# Two places exchange data between workgroups WITHOUT fences and are correct because of what the emitted instructions do
# on gfx950, not because of what the language promises (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement &
# inter-workgroup visibility": sc1 stores are written through, sc1 loads bypass the reader's L1 and are served by L2, a
# plain / sc0 store stays in the writer's XCD L2).  A compiler that changed the cache-policy bits of either access would
# change correctness, and the bit-equality GPU tests would only notice if a stale read happened to occur.  So the bits
# themselves are asserted here, on the code the Makefile's flags produce (hipcc cross-compiles without a GPU).
def _device_isa(name, tmp):
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not installed")
    out = os.path.join(str(tmp), name.replace(".hip", ".s"))
    import subprocess
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", out,
                    os.path.join(ROOT, "kccotgan_amd", "csrc", name)], check=True, stderr=subprocess.DEVNULL)
    kernels, cur, body = {}, None, []
    for line in open(out):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, body = m.group(1), []
            continue
        if cur is not None:
            if line.startswith(".Lfunc_end"):
                kernels[cur] = body
                cur = None
            else:
                body.append(line.strip())
    return kernels


def test_isa_of_the_multi_cu_sinkhorn_exchange(tmp_path):
    """sinkhorn_coop.hip, ll_store / ll_gather: every 64-bit {value, tag} exchange word is stored EITHER sc0
    (workgroup scope: stays in the XCD's L2, used only after the in-kernel placement check found every workgroup of the
    problem on one XCD) OR sc1 (agent scope: written through), never plain and never nt; every polled 64-bit load is sc1
    (L1 bypassed, L2-served); and the exchange itself holds no fence: the only write-back / invalidates of a kernel are the
    release / acquire of the one-time placement check (ll_same_xcd)."""
    kernels = _device_isa("sinkhorn_coop.hip", tmp_path)
    ll = {k: v for k, v in kernels.items() if "sinkhorn_fwd_ll" in k or "sinkhorn_bwd_ll" in k}
    assert len(ll) == 6, sorted(kernels)                           # fwd / bwd x EPT 4, 8, 16
    for name, body in ll.items():
        st = [l for l in body if l.startswith("global_store_dwordx2")]
        ld = [l for l in body if l.startswith("global_load_dwordx2")]
        flags = lambda l: set(l.split()[-2:]) & {"sc0", "sc1", "nt"}
        assert st and ld, name
        assert all(flags(l) in ({"sc0"}, {"sc1"}) for l in st), (name, st)
        n0, n1 = sum(flags(l) == {"sc0"} for l in st), sum(flags(l) == {"sc1"} for l in st)
        assert n0 == n1 >= 3, (name, n0, n1)                       # each ll_store site: one form per scope
        assert all(flags(l) == {"sc1"} for l in ld), (name, ld)
        # (the only flat_ accesses are the volatile reads of the LDS give-up flag; no exchange word travels through one)
        assert not any(l.startswith(("flat_load_dwordx2", "flat_store", "flat_atomic")) for l in body), name
        assert sum(l.startswith("buffer_wbl2") for l in body) == 1, name
        assert sum(l.startswith("buffer_inv") for l in body) == 2, name
    zero = [v for k, v in kernels.items() if "ll_zero" in k]
    assert len(zero) == 1 and any(l.startswith("global_store_dwordx2") and l.endswith("sc1") for l in zero[0])


def test_isa_of_the_three_cost_hand_off_to_the_combining_workgroup(tmp_path):
    """sinkhorn.hip, the mixed divergence's combine (gan_utils.py:225) by the last of three workgroups to take a ticket --
    the first row of the guide's table of measured hand-offs: the handed-off word is stored sc1 by ONE lane, that lane drains
    its stores (an asm s_waitcnt vmcnt(0) the compiler cannot drop), adds to the agent-scope ticket, and the workgroup whose
    add returned last reads the three words with sc1 loads after the add has returned.  gfx950 behaviour, not a language
    guarantee -- hence the pin."""
    kernels = _device_isa("sinkhorn.hip", tmp_path)
    seen = 0
    for name, body in kernels.items():
        for i, l in enumerate(body):
            if not l.startswith("global_atomic_add"):
                continue
            seen += 1
            assert l.split()[-1] == "sc0", (name, l)               # returning atomic (the ticket value decides who combines)
            back, ahead = body[max(0, i - 60):i], body[i:i + 30]
            drains = [j for j in range(len(back) - 1) if back[j].startswith(";;#ASMSTART") and back[j + 1] == "s_waitcnt vmcnt(0)"]
            assert drains, name
            stores = [j for j, b in enumerate(back) if b.startswith("global_store_dword ") and b.endswith("sc1")]
            assert stores and stores[-1] < drains[-1], (name, back)   # cost word written through BEFORE the drain
            loads = [b for b in ahead if b.startswith("global_load_dword ")]
            assert len(loads) == 3 and all(b.endswith("sc1") for b in loads), (name, loads)
            wait = [j for j, b in enumerate(ahead) if b.startswith("s_waitcnt vmcnt(0)")]
            first_load = ahead.index(loads[0])
            assert wait and wait[0] < first_load, name             # the add has returned before the first load issues
    assert seen >= 6, seen                                         # register forward kernels + fused kernels, every instantiation
