export REPS=5
for i in 1 2; do
DARK_AMD_LIB=$PWD/dark_amd/libdark_amd_base.so python tools/stage_time.py sa 1e8 text 2>&1 | grep SUMMARY | cut -c1-330
python tools/stage_time.py sa 1e8 text 2>&1 | grep SUMMARY | cut -c1-330
done
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
