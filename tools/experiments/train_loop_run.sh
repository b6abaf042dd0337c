#!/bin/bash
# The training loop end to end on the synthetic timestep dataset (device ray generation, schedules, pose feedback, logging,
# checkpoints), at the 4096-ray batch and at the reference's own 512 (configs/waymo.gin:17):
#   tools/experiments/train_loop_run.sh  ->  gpurun_out/train_loop.txt
cd /root/repo
out=gpurun_out/train_loop.txt
ver=$(python3 -c "from durf_amd import _lib; print(_lib.lib().durf_version())")
for bs in 4096 512; do
  echo "# python -m durf_amd.train_boxpose --gin_file configs/waymo.gin, batch $bs, 3000 steps, synthetic timestep dataset (MI355X), library version $ver" >> $out
  rm -rf /tmp/tl_run
  python3 -m durf_amd.train_boxpose --gin_file configs/waymo.gin --train_dir /tmp/tl_run --render_every 1500 \
     --gin_param "Config.batch_size = $bs" --gin_param "Config.max_steps = 3000" --gin_param "Config.print_every = 500" \
     --gin_param "Config.save_every = 3000" --gin_param "MipNerfModel.no_pose_opt = True" --gin_param "MipNerfModel.no_yaw_opt = True" 2>&1 | grep -v Warning | tail -12 >> $out
done
