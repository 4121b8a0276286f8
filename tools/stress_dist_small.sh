# the 'small' case of tests/test_gpu_dist_sequence.py on 4 ranks, repeatedly, until one run fails (its logs are printed).
# Before every run a process fills device memory with a byte pattern and exits, so that memory the workers allocate and
# do not initialise holds something else each time (POLLUTE=0 turns that off).

for it in $(seq 1 ${1:-10}); do
  if [ "${POLLUTE:-1}" = 1 ]; then
    python - <<PY
import torch
pat = [0xff, 0x7f, 0x01, 0x80, 0x3c, 0xc0][$it % 6]
xs = [torch.full((1 << 30,), pat, dtype=torch.uint8, device='cuda') for _ in range(48)]
torch.cuda.synchronize()
PY
  fi
  port=$((20000 + RANDOM % 20000))
  rm -f /tmp/stress_*.log
  pids=""
  for r in 0 1 2 3; do
    python tests/_sequence_worker.py small $r 4 $port /tmp/stress_out.npz > /tmp/stress_$r.log 2>&1 &
    pids="$pids $!"
  done
  fail=0
  for p in $pids; do wait $p || fail=1; done
  echo "iteration $it fail=$fail"
  if [ $fail = 1 ]; then for r in 0 1 2 3; do echo "--- rank $r"; grep -v "socket.cpp\|Gloo\|amdgpu.ids" /tmp/stress_$r.log | tail -70; done; break; fi
done
