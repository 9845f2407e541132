python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -5
export REPS=3
python tools/stage_time.py dc 268435456 acgt 2>&1 | grep -E "SUMMARY|Error|assert"
python tools/stage_time.py dc 1e8 text 2>&1 | grep -E "SUMMARY|Error|assert"
python tools/stage_time.py dc 1073741824 random 2>&1 | grep -E "SUMMARY|Error|assert"
