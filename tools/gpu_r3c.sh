#!/bin/bash
# round 3, GPU call C: the general tile kernels with the leaner sample bodies (forward 29 -> 23 VALU per sample, adjoint 34 -> 31):
# a small smoke first (bounded), tile-vs-ray-driven fuzz + parity tests, then A/B timings against the round-2 bodies
# (build/ab2/libtomo_clamp.so = the same sources with -DTOMO_TILE_CLAMP)
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 90 python3 tools/quick_bench.py fwd:128:8:tilt=1 adj:128:8:tilt=1 > $R/gpurun_out/r3c_smoke.log 2>&1 || { echo "smoke failed/timed out"; tail -5 $R/gpurun_out/r3c_smoke.log; exit 1; }
cat $R/gpurun_out/r3c_smoke.log
timeout -k 10 400 python3 -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q > $R/gpurun_out/r3c_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -E "passed|failed|FAILED|Error" $R/gpurun_out/r3c_pytest.log | tail -8
if [ $rc -ne 0 ]; then tail -60 $R/gpurun_out/r3c_pytest.log; exit $rc; fi
rm -f $R/gpurun_out/r3c_ab.log
for lib in "" "$R/build/ab2/libtomo_clamp.so"; do
  echo "== library: ${lib:-default}" | tee -a $R/gpurun_out/r3c_ab.log
  TOMO_AB_LIB=$lib timeout -k 10 300 python3 tools/quick_bench.py fwd:1024:64:tilt=1 adj:1024:64:tilt=1 fwd:1024:64:tilt=1:shepp=1 adj:512:96:tilt=2 fwd:512:96:tilt=2 2>&1 | tee -a $R/gpurun_out/r3c_ab.log
done
