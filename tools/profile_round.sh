#!/bin/bash
# Re-records profiles/rNN on the GPU box:  bash tools/profile_round.sh 2   (run through gpurun; writes gpurun_out/prof_rNN/)
# Every rocprofv3 pass profiles `python3 bench.py ...` with the launch policy FIXED (--launch-hint): the LIBRARY DEFAULT
# (hint 0) unless PROFILE_TUNED=1 asks for the policy the un-profiled line's autotune() kept on this box.  A profiled
# process makes one allocation and tries no placements, and a tuned policy can be placement-sensitive where the default is
# not (profiles/README.md, final_c: (5, 2) on C3 75.6 us tuned, 92 us in the profiled process); compare the --stats
# average with the line's roofline.frac_untuned_library.
R=${1:-2}
RR=$(printf "r%02d" $R)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$RR
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py"
WL=${WL:-"c3 c2 c5 v1 v2 v4 v5"}      # WL="v4 v5" re-records a subset
has() { case " $WL " in *" $1 "*) return 0;; esac; return 1; }
H=35
if has c3; then
$B > $OUT/bench_c3.json 2> $OUT/bench_c3.err || exit 1
H=$(python3 -c "import json; print(json.load(open('$OUT/bench_c3.json'))['config']['launch_hint'])")
[ -z "$PROFILE_TUNED" ] && H=0
echo "c3 launch hint $H"
$B --auto-reset --no-cpu-baseline > $OUT/bench_c3_autoreset.json 2>/dev/null
# kernel trace + stats, the timed policy on every launch
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3 -- $B --launch-hint $H --no-cpu-baseline > $OUT/bench_c3_under_rocprofv3.json 2> $OUT/stats_c3.err; echo "stats c3 rc=$?"
fi
for w in c2 c5 v1 v2 v4 v5; do has $w || continue; $B --workload $w > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "$w rc=$?"; done
# the launch policy each un-profiled line was tuned to on THIS box: fixed for every profiled pass below
hint() { [ -z "$PROFILE_TUNED" ] && { echo 0; return; }; python3 -c "import json; print(json.load(open('$OUT/bench_$1.json'))['config']['launch_hint'])" 2>/dev/null || echo 0; }
has c2 && $B --workload c2 --graph --no-cpu-baseline > $OUT/bench_c2_graph.json 2>/dev/null
for w in v1 v2 v4 v5 c5 c2; do
  has $w || continue
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$w -- $B --workload $w --launch-hint $(hint $w) --no-cpu-baseline > $OUT/bench_${w}_under_rocprofv3.json 2> $OUT/stats_$w.err; echo "stats $w rc=$?"
done
# PMC passes (separate: WRITE_SIZE and FETCH_SIZE do not fit the TCC slots together)
for w in c3 v1 v2 v4 v5 c5; do
  has $w || continue
  X="--workload $w --launch-hint $(hint $w)"; [ $w = c3 ] && X="--launch-hint $H"
  for c in WRITE_SIZE FETCH_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${w}_$c -- $B $X --steps 20 --warmup 5 --no-cpu-baseline > $OUT/pmc_${w}_$c.json 2> $OUT/pmc_${w}_$c.err; echo "pmc $w $c rc=$?"
  done
done
for w in c3 v2 v4 v5; do
  has $w || continue
  X="--workload $w --launch-hint $(hint $w)"; [ $w = c3 ] && X="--launch-hint $H"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/sq_$w -- $B $X --steps 20 --warmup 5 --no-cpu-baseline > $OUT/sq_$w.json 2> $OUT/sq_$w.err; echo "sq $w rc=$?"
done
# keep the merge small: summaries only (the raw counter files are tens of MB)
cd $ROOT
python3 tools/profile_summarise.py $OUT $R > $OUT/summary.log 2>&1; echo "summarise rc=$?"; tail -40 $OUT/summary.log
find $OUT -name "*counter_collection.csv" -size +3M -delete; find $OUT -name "*kernel_trace.csv" -size +3M -delete
du -sh $OUT
