for prec in bf16 split; do
rm -rf /tmp/run_$prec
( time python -m outdoor_nerf_depth_amd.ddp_train_nerf --synthetic --synthetic_frames 60 --expname kitti_shaped --basedir /tmp/run_$prec \
    --use_depth --depth_sup_type gt --depth_loss_type mse --lambda_depth 0.1 --cascade_samples 64,128 \
    --N_iters 5001 --i_print 1000 --i_weights 5000 --i_test 5000 --testskip 2 --precision $prec --world_size 1 ) 2>&1 | grep -E "step: |test_|real" | sed -e 's/level_0\/loss_depth.*level_1\/rgb_loss/... level_1\/rgb_loss/' 
ls /tmp/run_$prec/kitti_shaped/render_test_005000 | tr '\n' ' '; echo
done
