cd $GRAFT_REPO_ROOT
python -m pytest tests/test_split_storage_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -4
python -m pytest tests/test_phiseg_gpu.py -x -q -p no:cacheprovider -s -k trajectory 2>&1 | grep -E "^step|passed|failed|Error" | head
python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=1800 -x 2>&1 | tail -6
