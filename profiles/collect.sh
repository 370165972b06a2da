#!/bin/bash
# Run on the GPU box from the repo root:  bash profiles/collect.sh <tag>   (outputs under gpurun_out/prof_<tag>/)
# Passes: kernel trace + stats; FETCH_SIZE; WRITE_SIZE; SQ instruction mix.  --pmc is never combined with trace flags.
set -e
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-other --no-multi > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-other --no-multi > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-other --no-multi > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY --output-format csv -d $O/sq -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-other --no-multi > $O/sq.log 2>&1
# second SQ pass (8 slots per pass): where waves are parked, LDS behaviour, vector-memory instruction mix
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $O/sq2 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-other --no-multi > $O/sq2.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sq3 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-other --no-multi > $O/sq3.log 2>&1 || true
# a reference-written frame (oracle encoder) through the token discovery + symbolic decoder: kernel stats of 5 decodes
rocprofv3 --kernel-trace --stats --output-format csv -d $O/reftrace -- python3 $R/tools/region_debug.py --mib 1024 --dataset f32 --reps 5 > $O/reftrace.log 2>&1 || true
cp $(ls $O/reftrace/*/*kernel_stats.csv | head -1) $O/reference_frame_kernel_stats.csv || true
rocprofv3 --kernel-trace --stats --output-format csv -d $O/owntrace -- python3 $R/tools/region_debug.py --mib 1024 --dataset f32 --reps 5 --writer device > $O/owntrace.log 2>&1 || true
cp $(ls $O/owntrace/*/*kernel_stats.csv | head -1) $O/indexless_frame_kernel_stats.csv || true
# the same data as a foreign Snappy frame (oracle's 64 KiB-block encoder): element discovery (k_snr_*) + symbolic decode, kernel stats of 3 decodes
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sntrace -- python3 $R/tools/region_debug.py --codec snappy --mib 1024 --dataset f32 --reps 3 > $O/sntrace.log 2>&1 || true
cp $(ls $O/sntrace/*/*kernel_stats.csv | head -1) $O/snappy_foreign_frame_kernel_stats.csv || true
# SQ counters of the same reference-written frame's decode (token discovery k_rg_*, symbolic decode k_sy_*): instruction mix, waits, waves
A="$R/tools/region_debug.py --mib 1024 --dataset f32 --reps 2"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY --output-format csv -d $O/refsq -- python3 $A > $O/refsq.log 2>&1 || true
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $O/refsq2 -- python3 $A > $O/refsq2.log 2>&1 || true
(cd $R && python3 profiles/summarize.py counters $O/refsq $O/reference_frame_sq.csv && python3 profiles/summarize.py counters $O/refsq2 $O/reference_frame_sq2.csv) || true
# everything bench.py measures beside the headline (configs 3 / 4 / 5, index-less and reference-written decodes, C-Blosc-1 frames, the
# small-frame batches): one kernel trace of a short full run
rocprofv3 --kernel-trace --stats --output-format csv -d $O/alltrace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-multi > $O/alltrace.log 2>&1 || true
cp $(ls $O/alltrace/*/*kernel_stats.csv | head -1) $O/all_configs_kernel_stats.csv || true
cd $R
# config 2 (filters only): kernel stats + HBM traffic of the shuffle / unshuffle kernels
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ftrace -- python3 $R/bench.py --mode filter --steps 5 --warmup 2 --no-cpu --no-multi > $O/ftrace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/ffetch -- python3 $R/bench.py --mode filter --steps 3 --warmup 1 --no-cpu --no-multi > $O/ffetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/fwrite -- python3 $R/bench.py --mode filter --steps 3 --warmup 1 --no-cpu --no-multi > $O/fwrite.log 2>&1
python3 profiles/summarize.py traffic $O/ffetch $O/fwrite $O/filter_traffic.json "bench.py --mode filter: 1024 MiB D-f32, shuffle + unshuffle typesize 4, 1 GPU (MI355X)"
cp $(ls $O/ftrace/*/*kernel_stats.csv | head -1) $O/filter_kernel_stats.csv
python3 profiles/summarize.py traffic $O/fetch $O/write $O/traffic.json "bench.py default: 1024 MiB D-f32 frame, Shuffle1 ts=4 + LZ4, index trailer, 1 GPU (MI355X)"
python3 profiles/summarize.py counters $O/sq $O/sq.csv
python3 profiles/summarize.py counters $O/sq2 $O/sq2.csv
python3 profiles/summarize.py counters $O/sq3 $O/sq3.csv || true
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
tail -1 $O/trace.log | cut -c1-300
