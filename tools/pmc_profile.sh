#!/bin/bash
# PMC passes for the bench workload (each counter set in its own run, --kernel-trace only; see MI355X_MICROARCH.md
# "rocprofv3 PMC slots"): HBM read bytes, HBM write bytes, MFMA busy / issue counters, wave-state counters.
# Usage (on the GPU box): [PMC_BENCH_ARGS="--precision f64"] tools/pmc_profile.sh <outdir> [pass ...]      (no pass names: the standard set; "icache", "issue": diagnostics)
set -e
OUT=${1:-gpurun_out/pmc}
shift || true
ONLY=" $* "
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
run() { name=$1; shift; if [ "$ONLY" != "  " ] && [[ "$ONLY" != *" $name "* ]]; then return 0; fi; rocprofv3 --kernel-trace --pmc "$@" -d $REPO/$OUT/$name -o p -- python $REPO/bench.py --steps ${PMC_STEPS:-6} --warmup ${PMC_WARMUP:-2} --no-cpu-baseline $PMC_BENCH_ARGS > $REPO/$OUT/$name.json 2> $REPO/$OUT/$name.err || echo "pass $name failed"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run mfma_bf16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA      # split-operand GEMMs: the share of the matrix work on the bf16 pipe
run waves SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS
run l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum        # is the weight stream (packed image re-read by every workgroup) served by the L2?
if [ "$ONLY" != "  " ]; then
# diagnostics, on request only: instruction-cache behaviour of the (large, fully unrolled) fused kernel; issue-side busy / wait split
run icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH
run issue SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES
fi
ls -R $REPO/$OUT | head -30
# text summary per pass + profiles/pmc_traffic.json (HBM-side bytes per launch of the fused kernels; read by bench.py's roofline.traffic)
python $REPO/tools/pmc_summarize.py $REPO/$OUT $REPO/$OUT/pmc_summary.txt $REPO/$OUT/pmc_traffic.json
# (the per-pass databases of a long workload exceed what gpurun merges back: the text summary is what is kept)
if [ -n "$PMC_DROP_DB" ]; then find $REPO/$OUT -name '*.db' -delete; fi
