#!/bin/bash
# does any runtime knob move the 1.67 us dependent-launch floor?  (first 3 lines of the launch-floor micro-benchmark)
run() { echo "== $*"; env "$@" timeout 120 tools/micro/launch_floor 2>&1 | sed -n 2,4p; }
run X=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run GPU_MAX_HW_QUEUES=1
run HSA_ENABLE_INTERRUPT=0
run HIP_FORCE_DEV_KERNARG=1 HSA_ENABLE_INTERRUPT=0
run AMD_DIRECT_DISPATCH=0
