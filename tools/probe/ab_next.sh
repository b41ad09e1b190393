for e in "A=1" "HVR_FRAME_GROUPS=2" "HVR_FUSE_NEXT=0" "HVR_FUSE_TAIL=0 HVR_FUSE_NEXT=0"; do
echo "== $e"; env $e python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "bf16_training_step" 2>&1 | grep -o "q_data_fc_1.weight', [0-9.]*, [0-9.]*\|passed\|failed" | head -3
done
