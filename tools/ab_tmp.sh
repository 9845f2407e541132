python -m pytest tests/test_gpu_parity.py tests/test_env_variants.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3
export REPS=3
python tools/stage_time.py sa 1e8 text 2>&1 | grep SUMMARY
python tools/stage_time.py sa 1073741824 random 2>&1 | grep SUMMARY
export DARK_AMD_LIB=$PWD/dark_amd/libdark_amd_tuning.so
for d in 0 1; do DK_DIGIT_PLANE=$d python tools/stage_time.py sa 1e8 text 2>&1 | grep SUMMARY; done
for d in 0 1; do DK_DIGIT_PLANE=$d python tools/stage_time.py sa 268435456 acgt 2>&1 | grep SUMMARY; done
