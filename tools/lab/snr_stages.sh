# chain state of a Snappy block after each stage of the discovery (lab): bash tools/lab/snr_stages.sh [mib]
for k in 0 1 2 3 4; do echo "== stop after stage $k"; HIPBLOSC_DEBUG_SNR_STOP=$k timeout -k 10 120 python tools/region_debug.py --codec snappy --mib ${1:-256} --reps 1 2>&1 | grep -E "^needfull" | cut -c1-260; done
