bash tools/ab_opts.sh 3 "wgrad_wino_rounds=2" "wgrad_wino_rounds=1"
for i in 1 2; do for r in 3 2; do echo -n "MRI VNet rounds=$r: "; python tools/bench_workloads.py --model VNet --shape 512,512,12 --num-classes 20 --batch 1 --steps 8 --opt wgrad_wino_rounds=$r 2>&1 | grep -E "ms/step" | tail -1; done; done
for r in 3 2; do echo -n "UNet3D fp16 rounds=$r: "; python tools/bench_workloads.py --model UNet3D --precision fp16 --shape 192,192,64 --num-classes 3 --batch 2 --opt wgrad_wino_rounds=$r 2>&1 | grep -E "ms/step" | tail -1; done
