export REPS=3
for a in "1e8 text" "268435456 acgt" "1073741824 random"; do python tools/stage_time.py sa $a 2>&1 | grep SUMMARY; done
python tools/pathological.py full gpurun_out/r3_patho_full.json
