python -m pytest tests/test_gpu_unet3d.py tests/test_gpu_fullsize.py -q -s -k "unet3d" 2>&1 | grep -E "UNet3D|passed|failed|Error|assert" | head -20
