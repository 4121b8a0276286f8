# Every differential fuzzer and stress tool once, with a seed (default 1): about a minute on one MI355X.
# usage: bash tools/fuzz_all.sh [seed]     (on the GPU box: gpurun -- 'bash tools/fuzz_all.sh 7')
S=${1:-1}
cd "$(dirname "$0")/.."
rc=0
run() { echo "== $*"; timeout 600 "$@" < /dev/null 2>&1 | tail -1; [ ${PIPESTATUS[0]} -eq 0 ] || rc=1; }
run python tools/fuzz_frames.py 800 $S
run python tools/fuzz_dirs.py 400 $S
PADDED=1 run python tools/fuzz_frames.py 400 $S
PADDED=1 run python tools/fuzz_dirs.py 200 $S
BIG=5 run python tools/fuzz_dirs.py 60 $S
run python tools/fuzz_widened.py 800 $S
run python tools/fuzz_ops.py 800 $S
run python tools/fuzz_sequence.py 80 $S
RESIDENT=1 run python tools/fuzz_sequence.py 80 $S
PINNED=1 run python tools/fuzz_sequence.py 80 $S
run python tools/fuzz_mapping.py 800 $S
run python tools/fuzz_cubic.py 100 $S
METHOD=linear run python tools/fuzz_cubic.py 100 $S
PIN=1 run python tools/stress_sequence.py 300 530 354
run python tools/stress_params.py
exit $rc
