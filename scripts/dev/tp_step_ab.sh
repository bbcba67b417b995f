#!/bin/bash
# two-phase time step with the recompute-state mode (default) and with the streamed state (kernel variant 4)
python3 scripts/time_two_phase.py 64 4 2>&1 | grep "step "
ADAFLO_KERNEL_VARIANT=4 python3 scripts/time_two_phase.py 64 4 2>&1 | grep "step "
