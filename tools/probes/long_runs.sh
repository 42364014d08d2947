for cfg in "bf16 20001 gt mse 0.1" "fp16_fwd 5001 mono_crop kl 0.1" "split_fwd 5001 stereo_crop l1 1.0"; do
set -- $cfg
rm -rf /tmp/run_long
python -m outdoor_nerf_depth_amd.ddp_train_nerf --synthetic --synthetic_frames 60 --expname long --basedir /tmp/run_long \
    --use_depth --depth_sup_type $3 --depth_loss_type $4 --lambda_depth $5 --cascade_samples 64,128 \
    --N_iters $2 --i_print 5000 --i_weights 100000 --i_test $(( $2 - 1 )) --testskip 3 --precision $1 --world_size 1 2>&1 | grep -E "step: [0-9]*000 |test_psnr" | sed -e 's/level_0\/loss_depth.*level_1\/rgb_loss/... level_1\/rgb_loss/' | cut -c1-200
echo "== $cfg done"
done
