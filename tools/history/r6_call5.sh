#!/bin/bash
# Round 6, call 5: bisecting what costs the submit / wait pipeline its overlap (tools/history/r6_pipeline_ab.py with DEMI_TICKET_BISECT and
# the same sequence enqueued on two torch streams)
mkdir -p gpurun_out
python __graft_entry__.py > gpurun_out/r06_call5_build.log 2>&1
timeout 900 python tools/history/r6_pipeline_ab.py > gpurun_out/r06_call5_pipeline_ab.txt 2>&1
cat gpurun_out/r06_call5_pipeline_ab.txt
