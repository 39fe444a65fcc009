#!/bin/bash
# Everything profiles/ holds for the current tree, in one GPU call (run on the GPU box from the repo root):
# rocprofv3 stats of the three bench lines, FETCH/WRITE PMC passes, the un-profiled bench lines, per-step times,
# the single-utterance latency breakdown.  Outputs under gpurun_out/refresh/; copy what is to be kept to profiles/.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
bash $R/tools/prof_round.sh
D=$O/refresh
rm -rf $D && mkdir -p $D
cd $R
A="HBM-side bytes per launch, rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of tools/profile_steps.py (6 identical passes of 64 units, full model, fp32). Counters are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads); WRITE_SIZE as read. Includes per-lane scratch (register spill) traffic."
python3 tools/pmc_traffic.py $O/prof_pmc_f32_fetch/fetch_counter_collection.csv $O/prof_pmc_f32_write/write_counter_collection.csv $D/r01_pmc_traffic_g64.json 64 "$A" > $D/r01_pmc_traffic_g64.txt
python3 tools/pmc_traffic.py $O/prof_pmc_bf16_fetch/fetch_counter_collection.csv $O/prof_pmc_bf16_write/write_counter_collection.csv $D/r01_pmc_traffic_bf16_g64.json 64 "As r01_pmc_traffic_g64.json, for tools/bf16_steps.py (bf16 conv-GEMM mode with the packed activation exchange, 6 passes of 64 units)." > $D/r01_pmc_traffic_bf16_g64.txt
python3 tools/pmc_traffic.py $O/prof_pmc_enc_fetch/fetch_counter_collection.csv $O/prof_pmc_enc_write/write_counter_collection.csv $D/r01_encoder_pmc.json 64 "As r01_pmc_traffic_g64.json, for tools/enc_steps.py (encoder path, 6 passes of 64 clips x 10 s)." > $D/r01_encoder_pmc.txt
cp $O/prof_f32/f32_kernel_stats.csv $D/r01_kernel_stats_bench_g64.csv
cp $O/prof_bf16/bf16_kernel_stats.csv $D/r01_kernel_stats_bench_bf16_g64.csv
cp $O/prof_enc/enc_kernel_stats.csv $D/r01_kernel_stats_bench_encoder.csv
cp $O/bench_under_rocprof_f32.json $D/r01_bench_under_rocprof_g64.json
cp $O/bench_under_rocprof_bf16.json $D/r01_bench_under_rocprof_bf16_g64.json
cp $O/bench_under_rocprof_enc.json $D/r01_bench_under_rocprof_encoder.json
rm -rf $O/prof_*
# the PMC JSONs the benches read are the fresh ones
cp $D/r01_pmc_traffic_g64.json $D/r01_pmc_traffic_bf16_g64.json $D/r01_encoder_pmc.json $R/profiles/
timeout -k 10 300 python3 bench.py > $D/r01_bench_g64.json 2> $D/bench_f32.err
timeout -k 10 300 python3 bench.py --dtype bf16 > $D/r01_bench_bf16_g64.json 2> $D/bench_bf16.err
timeout -k 10 300 python3 bench_encoder.py > $D/r01_bench_encoder.json 2> $D/bench_enc.err
timeout -k 10 200 python3 tools/profile_steps.py > $D/r01_steps_g64_v7.txt 2>&1
timeout -k 10 200 python3 tools/bf16_steps.py > $D/r01_steps_bf16_g64.txt 2>&1
timeout -k 10 200 python3 tools/enc_steps.py > $D/r01_steps_encoder.txt 2>&1
timeout -k 10 200 python3 tools/latency_breakdown.py > $D/r01_latency_single_utterance.txt 2>&1
echo refresh done
