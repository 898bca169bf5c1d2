#!/bin/bash
# round 5, call 28: the GPU suite of the FINAL binary after the test-only fix of call 27's three failures (a property called as a function in the new test)
mkdir -p gpurun_out/r05x
python -m pytest tests -m gpu -q --timeout 1500 2>&1 | tail -8 > gpurun_out/r05x/suite.log
tail -3 gpurun_out/r05x/suite.log
