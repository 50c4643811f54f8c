#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel-trace stats + the PMC passes of the bench command for one config.
#   tools/collect_profiles.sh cfg2 [extra bench args]
# Writes under gpurun_out/prof_<cfg>/ and the summary gpurun_out/${ROUND}_pmc_<cfg>_summary.json; copy what is to be judged into profiles/.
set -u
CFG=${1:-cfg2}; shift || true
ROUND=${ROUND:-r06}
DEFB=4096; [ "$CFG" = cfg4 ] && DEFB=8192      # (per-GPU batch of the config: bench.py matches the summary on it)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$CFG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-extras $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $BENCH > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > /dev/null 2> $OUT/pmc_write.err
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $ROOT/gpurun_out/${ROUND}_bench_${CFG}_kernel_stats.csv
# (the source fingerprint beside the CSV: bench.py quotes the rocprof averages in its line only for a summary of ITS kernel sources)
(cd $ROOT && python3 -c "import json,sys; sys.path.insert(0,'.'); from ncde_amd import _lib; json.dump({'source_fingerprint': _lib.source_fingerprint(), 'command': '$BENCH'}, open('gpurun_out/${ROUND}_bench_${CFG}_kernel_stats.meta.json','w'))")
cp $OUT/bench_trace.json $ROOT/gpurun_out/${ROUND}_bench_${CFG}_profiled.json 2>/dev/null
cd $ROOT && python3 tools/pmc_summary.py $OUT $ROOT/gpurun_out/${ROUND}_pmc_${CFG}_summary.json --config $CFG --batch ${PMC_BATCH:-$DEFB} --passes 8 --note "rocprofv3 --kernel-trace --pmc, three separate passes (SQ counters | FETCH_SIZE | WRITE_SIZE) of: $BENCH" > /dev/null
tail -c 400 $OUT/bench_trace.json; echo; head -6 $ROOT/gpurun_out/${ROUND}_bench_${CFG}_kernel_stats.csv | cut -c1-160
