#!/usr/bin/env bash
# third GPU call of round 6: A/B of the Horner forms, the full GPU suite on the new build, multinomial p2p, IPC connect probe
set -u
OUT=gpurun_out/r06c
mkdir -p $OUT
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 bash tools/ab_bench.sh mcl r05 plain default > $OUT/ab_mcl.txt 2>&1; cat $OUT/ab_mcl.txt | tee -a $OUT/summary.txt
timeout 500 bash tools/ab_bench.sh fastslam r05 plain default > $OUT/ab_fs1.txt 2>&1; cat $OUT/ab_fs1.txt | tee -a $OUT/summary.txt
timeout 400 bash tools/ab_bench.sh fastslam2 plain default > $OUT/ab_fs2.txt 2>&1; cat $OUT/ab_fs2.txt | tee -a $OUT/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -8 $OUT/pytest_gpu.txt | tee -a $OUT/summary.txt
L="python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1"
OMP_NUM_THREADS=1 RR_P2P_CU_PARTITION=1 RR_WORKER_PEAKED=32 timeout 400 $L --master-port 29771 tests/_gpu_p2p_worker.py 250000 12 > $OUT/ipc_mcl_250k_partition.out 2> $OUT/ipc_mcl_250k_partition.err; echo "ipc_mcl_250k_partition rc=$? ok=$(grep -o P2P_OK $OUT/ipc_mcl_250k_partition.out | wc -l) $(grep P2P_TOPOLOGY $OUT/ipc_mcl_250k_partition.out)" | tee -a $OUT/summary.txt
RR_BENCH_SHARE_DEVICE=1 RR_BENCH_DEADLINE_S=500 RR_P2P_CU_PARTITION=1 timeout 600 python bench.py --gpus 8 --steps 20 --warmup 5 --particles 250000 --no-extra-legs --no-cpu-baseline > $OUT/bench8_250k.out 2> $OUT/bench8_250k.err; echo "bench8_250k rc=$?" | tee -a $OUT/summary.txt
grep -h "VALIDATION\|validated\|failed validation\|gave up\|timed out" $OUT/bench8_250k.err | head -20 | tee -a $OUT/summary.txt
tail -1 $OUT/bench8_250k.out | cut -c1-600 | tee -a $OUT/summary.txt
for spec in "2 125000 200" "8 125000 50" "8 125000 100" "8 125000 200"; do set -- $spec; RR_PROBE_DUMP_S=120 OMP_NUM_THREADS=1 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$1 --master-addr 127.0.0.1 --master-port 2978$1 tools/ipc_connect_probe.py $2 $3 > $OUT/ipc_probe_$1_$2_$3.out 2> $OUT/ipc_probe_$1_$2_$3.err; echo "ipc_probe $spec rc=$?: $(grep IPC_PROBE $OUT/ipc_probe_$1_$2_$3.out | head -2 | tr '\n' ' ')" | tee -a $OUT/summary.txt; done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err; echo "bench default rc=$?" | tee -a $OUT/summary.txt
cp bench_legs.json $OUT/bench_driver_command_legs.json 2>/dev/null
tail -1 $OUT/bench_driver_command.json | cut -c1-1500 | tee -a $OUT/summary.txt
