#!/bin/bash
# ablation: the round-3 flat forward without its float atomics (-DTOMO_ABLATE_FWD_ATOMICS=1); the plain-store variant (=2) existed until the
# aligned-window row tail and is measured in profiles/round3_fwd_tab_variants.md
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -f $R/gpurun_out/r3l.log
for lib in "" "$R/build/ab2/libtomo_noatomic1.so"; do
  echo "== library: ${lib:-default}" | tee -a $R/gpurun_out/r3l.log
  TOMO_AB_LIB=$lib timeout -k 10 300 python3 tools/quick_bench.py fwd:1024:128:tilt=0 fwd:1024:128:tilt=0:shepp=1 2>&1 | tee -a $R/gpurun_out/r3l.log
done
