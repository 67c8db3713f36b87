#!/bin/bash
# round 6: trace + PMC passes of the kernels that changed this round, each on its own workload (through gpurun from the repo root):
#   tools/prof_r6_kernels.sh   -> gpurun_out/r6k_<leg>_{trace,fetch,write,sq,mfma}/..., summaries printed
set -e
for leg in "gmm tools/ablate/run_gmm.py" "gen tools/ablate/run_gen_joint.py gen" "mdtril tools/ablate/run_md_tril.py" "met2m tools/ablate/run_metrics_one.py 1000000" "met20k tools/ablate/run_metrics_one.py 10000"; do
  set -- $leg; name=$1; shift
  GMM_DENSE=0 tools/prof_cmd.sh r6k_${name}_trace "$@" | head -8
  GMM_DENSE=0 tools/pmc_cmd.sh r6k_${name}_fetch "FETCH_SIZE" "$@" | grep -v "^$" | head -8
  GMM_DENSE=0 tools/pmc_cmd.sh r6k_${name}_write "WRITE_SIZE" "$@" | grep -v "^$" | head -8
  GMM_DENSE=0 tools/pmc_cmd.sh r6k_${name}_sq "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY" "$@" | grep -v "^$" | head -8
  GMM_DENSE=0 tools/pmc_cmd.sh r6k_${name}_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "$@" | grep -v "^$" | head -8
done
