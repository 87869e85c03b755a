#!/bin/bash
# A/B reference: build kccotgan_amd/csrc/libkccot_old.so from the csrc/ sources of another commit (default HEAD~1),
# compiled out of a scratch copy under /tmp so that the working tree is untouched.  Used by tools/ab_tile.sh,
# tools/ab_apply.sh and tools/prof_ab.sh ("old" = that library, "new" = the working tree's libkccot.so).
# usage: tools/build_old_lib.sh [commit]
set -e
REV=${1:-HEAD~1}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
S=/tmp/kccot_old_src
rm -rf $S && mkdir -p $S/a/b $S/include
git -C "$ROOT" archive "$REV" kccotgan_amd/csrc include | tar -x -C $S/a/b --strip-components=0
cp $S/a/b/include/kccot.h $S/include/
cd $S/a/b/kccotgan_amd/csrc 2>/dev/null || { echo "no csrc in $REV"; exit 1; }
# common.h includes ../../include/kccot.h relative to csrc/
mkdir -p ../../include && cp $S/include/kccot.h ../../include/
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -shared -o "$ROOT/kccotgan_amd/csrc/libkccot_old.so" *.hip
ls -la "$ROOT/kccotgan_amd/csrc/libkccot_old.so"
