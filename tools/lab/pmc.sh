set -e
R=$PWD; O=$R/gpurun_out/pmc_$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY --output-format csv -d $O/sq -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-other --no-multi > $O/sq.log 2>&1
cd $R && python3 profiles/summarize.py counters $O/sq $O/sq.csv && grep -E "kernel|k_match_fused|k_dec_indexed|k_stitch" $O/sq.csv
