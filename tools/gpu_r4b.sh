#!/bin/bash
tag=$1
PFP_TEST_HOOKS=1 PFP_PARSE_REC=1 PFP_VERBOSE=3 timeout -k 10 120 python tools/rec_check.py gpu > gpurun_out/${tag}_rec_dbg.log 2>&1
echo rc=$?
grep -v "^\[pfbwt_hip\] launch\|^\[pfbwt_hip\] suffix\|K=" gpurun_out/${tag}_rec_dbg.log | head -30
grep "launch" gpurun_out/${tag}_rec_dbg.log | tail -5
