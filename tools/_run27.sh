cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -rf /tmp/out_train
timeout 900 python train.py --config configs/synthetic/vnet_synthetic_ct_128.yml --iters 20 --log_iters 5 --save_interval 10 --do_eval --save_dir /tmp/out_train > gpurun_out/r27_train.log 2>&1
ls -R /tmp/out_train | head -30 >> gpurun_out/r27_train.log
timeout 600 python val.py --config configs/synthetic/vnet_synthetic_ct_128.yml --model_path /tmp/out_train/best_model/model.pdparams --save_dir /tmp/out_val --auc_roc 1 > gpurun_out/r27_val.log 2>&1
ls -la /tmp/out_val | head >> gpurun_out/r27_val.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r27_smoke.log 2>&1
