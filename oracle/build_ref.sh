#!/usr/bin/env bash
# TEST INFRASTRUCTURE - builds the *untouched* reference Fortran where it lies
# under /root/reference into oracle/_ref/ (git-ignored, so the history stays source-only; NOT gpurun-ignored since round 3:
# the built binaries travel to the GPU box like our own .so files, where bench.py times libref_mf.so as `cpu_baseline.reference`).
# Nothing from the reference is copied into the repo; only compiled objects and
# f2py-GENERATED wrapper sources land in oracle/_ref/.
#
#   oracle/_ref/libref_mf.so        matrix-free float32 routines: forward_project_,
#                                   back_project_, compute_gradient_  (ctypes, all by-reference)
#   oracle/_ref/src/ray_wt_grad*.so f2py module (trilinear_ray_sparse / _interp)
#   oracle/_ref/src/vox_wt_grad*.so f2py module (bilinear_sparse / bilinear_vox_interp)
# The reference python packages are NOT linked or copied: tests/golden/make_golden.py puts
# oracle/_ref (for `src`) in front of /root/reference (for `utilities`, `recon`) on sys.path.
#
# Recipe follows the reference README.md:4-12 (gfortran/f2py3) with AMD flang 22
# from ROCm 7.2 standing in for gfortran (the only Fortran compiler in the image).
set -euo pipefail
REF=${REF:-/root/reference}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/_ref"
FC=${FC:-/opt/rocm/lib/llvm/bin/flang}
if [ ! -d "$REF/src" ]; then echo "build_ref: $REF not present - skipping (GPU box uses prebuilt files)"; exit 0; fi
if [ ! -x "$FC" ]; then echo "build_ref: no flang at $FC - reference unbuildable here"; exit 3; fi
mkdir -p "$OUT/obj" "$OUT/src"
S="$REF/src"
# 1. matrix-free float32 set (A5, A6, A7 of SURVEY.md section 8a)
"$FC" -O2 -fPIC -shared -module-dir "$OUT/obj" -o "$OUT/libref_mf.so" \
    "$S/rotations_module.f90" "$S/external_forward_projection.f90" "$S/external_back_projection.f90" \
    "$S/forward_projection.f90" "$S/back_projection.f90" "$S/projection_gradient.f90"
# 2. f2py modules (A1, A2, A9): two-step build because numpy.distutils does not know AMD flang
PYINC=$(python3 -c "import sysconfig;print(sysconfig.get_paths()['include'])")
NPINC=$(python3 -c "import numpy;print(numpy.get_include())")
F2PYSRC=$(python3 -c "import numpy.f2py,os;print(os.path.join(os.path.dirname(numpy.f2py.__file__),'src'))")
EXT=$(python3 -c "import sysconfig;print(sysconfig.get_config_var('EXT_SUFFIX'))")
for m in ray_wt_grad vox_wt_grad; do
    W="$OUT/obj/$m"; mkdir -p "$W"
    (cd "$W" && python3 -m numpy.f2py "$S/$m.f90" -m "$m" --lower >/dev/null)
    gcc -O2 -fPIC -I"$PYINC" -I"$NPINC" -I"$F2PYSRC" -c "$W/${m}module.c" -o "$W/${m}module.o"
    gcc -O2 -fPIC -I"$PYINC" -I"$NPINC" -I"$F2PYSRC" -c "$F2PYSRC/fortranobject.c" -o "$W/fortranobject.o"
    "$FC" -O2 -fPIC -module-dir "$W" -c "$S/$m.f90" -o "$W/$m.o"
    WRAP=""
    if [ -f "$W/$m-f2pywrappers2.f90" ]; then "$FC" -O2 -fPIC -module-dir "$W" -c "$W/$m-f2pywrappers2.f90" -o "$W/wrap2.o"; WRAP="$W/wrap2.o"; fi
    "$FC" -shared -o "$OUT/src/$m$EXT" "$W/${m}module.o" "$W/fortranobject.o" "$W/$m.o" $WRAP
done
touch "$OUT/src/__init__.py"
rm -rf "$OUT/obj"
echo "build_ref: ok -> $OUT"
