#!/bin/bash
# A/B of the hdemucs_mmi fp16 step under switches, alternating in one process group: bash tools/micro/hdemucs_ab.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2; do
  echo "--- default"; python3 tools/micro/hdemucs_time.py f16 7
  echo "--- MI_LSTM_STEPS=1"; MI_LSTM_STEPS=1 python3 tools/micro/hdemucs_time.py f16 7
  echo "--- MI_NO_TAP_IMAGE=1"; MI_NO_TAP_IMAGE=1 python3 tools/micro/hdemucs_time.py f16 7
  echo "--- MI_H_LAST_TAP=1"; MI_H_LAST_TAP=1 python3 tools/micro/hdemucs_time.py f16 7
  echo "--- MI_H_TWO_STREAMS=1"; MI_H_TWO_STREAMS=1 python3 tools/micro/hdemucs_time.py f16 7
  echo "--- MI_NO_TAIL_OVERLAP=1"; MI_NO_TAIL_OVERLAP=1 python3 tools/micro/hdemucs_time.py f16 7
done
