#!/bin/bash
# per-phase clocks of a frame wavefront with LDS stamps (make dev-timing-lds): the latencies of the kernel as shipped
cd "$(dirname "$0")/.."
timeout -k 10 200 python tools/phase_clock.py --envs 4096 --steps 6 --multi 16 --frame --bench-actions 2>&1 | grep -v "launch "
