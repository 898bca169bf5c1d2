#!/bin/bash
# round 5, call 20: the tests that touch the streams changed in call 19 (DDP wrapper / train script over RCCL at world 1, the shard loader's copy stream)
python -m pytest tests/test_gpu_autograd.py tests/test_gpu_data.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3
