#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== zero-filled A"
MB_ZERO=1 tools/mb/mb_bp_sweep 64 8192 262144 0 108 111 120 108 111
echo "== clocks while running variant 111 for ~3 s"
(MB_REPS=1500 tools/mb/mb_bp_sweep 64 8192 262144 0 111 > /tmp/long.log 2>&1 &)
sleep 1.0
for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | head -6; sleep 0.4; done
sleep 2; cat /tmp/long.log
echo "== clocks while running variant 120 (no MFMA) for ~3 s"
(MB_REPS=2000 tools/mb/mb_bp_sweep 64 8192 262144 0 120 > /tmp/long2.log 2>&1 &)
sleep 1.0
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | head -6; sleep 0.4; done
sleep 2; cat /tmp/long2.log
