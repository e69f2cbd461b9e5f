#!/bin/bash
# Diagnostic build for scripts/luad_stamps.py: a copy of csrc with the stamps of the tile solvers switched off and finer
# stamps in the small-space kernels (small.h), -DMMHN_STAMPS -> build_ab/libsstamps.so (never shipped, never timed)
set -e
cd "$(dirname "$0")/.."
rm -rf build_ab/ssrc && mkdir -p build_ab/ssrc && cp metmhn_amd/csrc/* build_ab/ssrc/
sed -i 's/STAMP_FLUSH(TR ? 8 : 0);//' build_ab/ssrc/kernels.h build_ab/ssrc/wsolve.h build_ab/ssrc/tsolve.h
python3 - <<'PY'
p = 'build_ab/ssrc/small.h'
s = open(p).read()
def rep(a, b, cnt=1):
    global s
    assert s.count(a) == cnt, (s.count(a), a)
    s = s.replace(a, b)
rep("STAMP_FLUSH(0)", "STAMP_FLUSH(SPB == 64 ? 0 : 8)", 2)
rep("    STAMP(5);\n", "    STAMP(6);\n")
rep("    STAMP(4);\n", "    STAMP(5);\n")
rep("    STAMP(3);\n", "    STAMP(5);\n")
rep("  STAMP(2);\n", "  STAMP(4);\n")
rep("    STAMP(1);\n", "    STAMP(4);\n")
rep("    setup(dS[sp]);\n    STAMP(0);\n", "    setup(dS[sp]);\n")
rep("  STAMP_DECL;\n  STAMP_START;\n", "  STAMP_START;\n")
rep("  auto setup = [&](const Desc& dg) {\n    sync();", "  STAMP_DECL;\n  auto setup = [&](const Desc& dg) {\n    sync();\n    STAMP_START;")
rep("    sync();\n    const Desc& d = dsh;\n    const int k = d.k;\n    const Params<T>& P = par[d.pset];",
    "    sync();\n    STAMP(0);\n    const Desc& d = dsh;\n    const int k = d.k;\n    const Params<T>& P = par[d.pset];")
rep("    sync();\n    const int nL = 1 << nl;", "    sync();\n    STAMP(1);\n    const int nL = 1 << nl;")
rep("    sync();\n    const uint32_t V = 1u << k;\n    const T dmn = P.dm[n];", "    sync();\n    STAMP(2);\n    const uint32_t V = 1u << k;\n    const T dmn = P.dm[n];")
rep("      LID[x] = T(1) / (dob - dq);\n    }\n    sync();\n  };", "      LID[x] = T(1) / (dob - dq);\n    }\n    sync();\n    STAMP(3);\n  };")
open(p, 'w').write(s)
PY
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -Wno-comment -DMMHN_STAMPS -Iinclude -o build_ab/libsstamps.so build_ab/ssrc/engine.hip
ls -la build_ab/libsstamps.so
