python tools/bench_workloads.py --model UNet3D --shape 192,192,64 --num-classes 3 --batch 2 --precision fp16 --steps 5 2>&1 | tail -1
python tools/bench_workloads.py --model UNet3D --shape 192,192,64 --num-classes 3 --batch 2 --precision fp32 --steps 5 2>&1 | tail -1
ls tools/*.py | head -30
