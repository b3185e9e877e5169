#!/bin/bash
# GPU box: the randomised sweeps and long runs behind the parity claims, once more with the round's final kernels -> gpurun_out/soak_r6.txt
cd $GRAFT_REPO_ROOT
{
echo "== fuzz_icp 6000 (seeds 50000..)"; timeout 900 python3 tools/fuzz_icp.py 6000 50000 2>&1 | tail -2
echo "== fuzz_parity 1500 hard (seeds 70000..)"; timeout 900 python3 tools/fuzz_parity.py 1500 70000 hard 2>&1 | tail -2
echo "== fuzz_slam 300 (seeds 9000..)"; timeout 900 python3 tools/fuzz_slam.py 300 9000 2>&1 | tail -2
echo "== fuzz_slam 300 mode3 (seeds 9500..)"; timeout 900 python3 tools/fuzz_slam.py 300 9500 mode3 2>&1 | tail -2
echo "== fuzz_async 150 (seeds 4000..)"; timeout 900 python3 tools/fuzz_async.py 150 4000 2>&1 | tail -2
echo "== fuzz_batch 200 (seeds 6000..)"; timeout 900 python3 tools/fuzz_batch.py 200 6000 2>&1 | tail -2
echo "== fuzz_batch 150 mode3 (seeds 7000..)"; timeout 900 python3 tools/fuzz_batch.py 150 7000 mode3 2>&1 | tail -2
echo "== async_soak"; timeout 900 python3 tools/async_soak.py 2>&1 | tail -3
echo "== first_flip 600 cfg2"; timeout 1500 python3 tools/first_flip.py 600 cfg2 gpurun_out/first_flip_600.txt 2>&1 | tail -6
echo "== hand-offs under load (poses + grid digest must be equal line by line)"
for m in 0 3; do
  TSD_HALO_KERNEL=1 TSD_PDF_ARGMAX_KERNEL=1 timeout 600 python3 tools/handoff_stress.py 500 $m 2>&1 | tail -2
  timeout 600 python3 tools/handoff_stress.py 500 $m --load 2>&1 | tail -2
done
} > gpurun_out/soak_r6.txt 2>&1
tail -40 gpurun_out/soak_r6.txt
