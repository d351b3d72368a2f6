#!/bin/bash
# ONE parameterised script for the GPU calls of a round (replaces the per-call tools/r03_gpu_run*.sh wrappers):
#     gpurun --timeout S -- 'bash tools/gpu_call.sh <tag> <step> [<step> ...]'
# Everything lands in gpurun_out/<tag>/ (merged back by gpurun); summaries worth keeping are copied into profiles/ by hand afterwards.
# Steps:
#   tests            python -m pytest tests -m gpu -q                              -> tests.txt
#   tests:<expr>     ... -k "<expr>" -s                                             -> tests_<n>.txt
#   bench[:args]     python bench.py [args]                                         -> bench.json / bench.err
#   trace            rocprofv3 --kernel-trace --stats of bench.py                   -> trace/ + kernel_stats.txt
#   trace:<args>     the same with bench.py <args> (e.g. --precision f64)                        -> trace_<n>/ + kernel_stats_<n>.txt
#   pmc              tools/pmc_profile.sh (separate --pmc passes, kernel-trace only) -> pmc/
#   ab:<v1,v2,..>    tools/ab_compare.py over the variant libraries csrc/abl/libpinn_<v>.so (+ "head" = the product)   -> ab_<cfg>.txt
#   abcfg:<cfg>:<v1,v2,..>   the same on another config (cfg3, cfg4, cfg5)
#   abf64:<v1,v2,..> the same for the float64 evaluation mode of the bench workload (wall time per pinn_loss_grad_f64 call)   -> ab_f64.txt
#   configs          tools/bench_configs.py (all BASELINE configs, HIP events on)   -> all_configs.txt
#   proxy            tools/scaling_proxy.py                                         -> scaling_proxy.txt / .json
#   stamps:<lib>     tools/stamp_profile.sh with a PINN_STAMP build                 -> stamps.txt
#   py:<script args> python <script args>                                           -> py_<n>.txt
#   sh:<command>     bash -c "<command>" (environment variables in front of a tool)  -> sh_<n>.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
TAG=${1:?tag}; shift
O=gpurun_out/$TAG
mkdir -p "$O"
export TMPDIR=/tmp
export PINN_JIT_DIR="$PWD/$O/jit_cache"           # run-time specialised kernels of this call (kept out of the tree)
n=0
for step in "$@"; do
    n=$((n + 1))
    echo "=== [$TAG] step $n: $step ($(date +%T))"
    case "$step" in
        tests)      timeout 1500 python -m pytest tests -m gpu -q --durations=30 > "$O/tests.txt" 2>&1; tail -n 3 "$O/tests.txt" ;;
        tests:*)    timeout 1500 python -m pytest tests -m gpu -q -s -k "${step#tests:}" > "$O/tests_$n.txt" 2>&1; grep -v "^$" "$O/tests_$n.txt" | tail -n 60 ;;
        bench)      timeout 900 python bench.py > "$O/bench.json" 2> "$O/bench.err"; tail -c 1500 "$O/bench.json" ;;
        bench:*)    timeout 900 python bench.py ${step#bench:} > "$O/bench_$n.json" 2> "$O/bench_$n.err"; head -c 600 "$O/bench_$n.json"; echo ;;
        trace)      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$O/trace" -o t -- python "$OLDPWD/bench.py" --steps 80 --warmup 5 --no-cpu-baseline > "$OLDPWD/$O/bench_under_rocprof.json" 2> "$OLDPWD/$O/trace.err")
                    python profiles/rocpd_stats.py "$(find "$O/trace" -name '*_results.db' | head -n 1)" > "$O/kernel_stats.txt" 2>&1; head -n 12 "$O/kernel_stats.txt"; find "$O/trace" -name '*.db' -size +20M -delete ;;
        trace:*)    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$O/trace_$n" -o t -- python "$OLDPWD/bench.py" ${step#trace:} > "$OLDPWD/$O/bench_under_rocprof_$n.json" 2> "$OLDPWD/$O/trace_$n.err")
                    python profiles/rocpd_stats.py "$(find "$O/trace_$n" -name '*_results.db' | head -n 1)" > "$O/kernel_stats_$n.txt" 2>&1; head -n 16 "$O/kernel_stats_$n.txt"; find "$O/trace_$n" -name '*.db' -size +20M -delete ;;
        pmc)        timeout 1200 bash tools/pmc_profile.sh "$O/pmc" > "$O/pmc.log" 2>&1; tail -n 40 "$O/pmc/pmc_summary.txt" ;;
        ab:*)       timeout 900 python tools/ab_compare.py $(echo "${step#ab:}" | tr ',' ' ') > "$O/ab_cfg2.txt" 2>&1; grep -i "round\|rror\|median" "$O/ab_cfg2.txt" | cut -c1-200 ;;
        abf64:*)    timeout 900 python tools/ab_compare.py --precision f64 $(echo "${step#abf64:}" | tr ',' ' ') > "$O/ab_f64.txt" 2>&1; grep -i "round\|rror\|median" "$O/ab_f64.txt" | cut -c1-200 ;;
        abcfg:*)    rest=${step#abcfg:}; cfg=${rest%%:*}; timeout 900 python tools/ab_compare.py --cfg "$cfg" $(echo "${rest#*:}" | tr ',' ' ') > "$O/ab_$cfg.txt" 2>&1; grep -i "round\|rror\|median" "$O/ab_$cfg.txt" | cut -c1-200 ;;
        configs)    timeout 900 python tools/bench_configs.py > "$O/all_configs.txt" 2>&1; tail -n 30 "$O/all_configs.txt" ;;
        proxy)      timeout 1200 python tools/scaling_proxy.py --out "$O/scaling_proxy.json" > "$O/scaling_proxy.txt" 2>&1; tail -n 30 "$O/scaling_proxy.txt" ;;
        stamps:*)   timeout 600 bash tools/stamp_profile.sh "${step#stamps:}" > "$O/stamps.txt" 2>&1; tail -n 40 "$O/stamps.txt" ;;
        py:*)       timeout 1200 python ${step#py:} > "$O/py_$n.txt" 2>&1; tail -n 40 "$O/py_$n.txt" ;;
        sh:*)       timeout 1200 bash -c "${step#sh:}" > "$O/sh_$n.txt" 2>&1; tail -n 40 "$O/sh_$n.txt" ;;
        *)          echo "unknown step $step" ;;
    esac
done
rm -rf "$O/jit_cache"/*/*/src 2>/dev/null
du -sh "$O" | cut -f1
