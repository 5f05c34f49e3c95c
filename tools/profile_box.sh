#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel-trace stats + separate PMC passes for one bench config.
# usage: tools/profile_box.sh <config> <steps> <tag> [f32]
set -o pipefail
CFG=${1:-2}; STEPS=${2:-200}; TAG=${3:-r02}; F32=${4:-}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
EXTRA="--no-cpu --no-extra --no-pmc --no-audition --repeats 2"  # as-allocated ring: every launch of the run writes the same buffers
SUF=""
if [ "$F32" = "f32" ]; then EXTRA="$EXTRA --obs-f32"; SUF="_f32"; fi
OUT=$ROOT/gpurun_out/prof_${TAG}_c${CFG}${SUF}
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --config $CFG --steps $STEPS --warmup 20 $EXTRA --detail $OUT/bench_detail.json > $OUT/bench_trace.json 2> $OUT/trace.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --config $CFG --steps 12 --warmup 4 $EXTRA > $OUT/bench_fetch.json 2> $OUT/fetch.err || exit 2
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --config $CFG --steps 12 --warmup 4 $EXTRA > $OUT/bench_write.json 2> $OUT/write.err || exit 3
echo "profiled config $CFG -> $OUT"
