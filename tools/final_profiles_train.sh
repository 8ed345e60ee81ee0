R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final2; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 80 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o p -- python $R/tools/train_bench.py --batch 64 --steps 3 > $O/prof_train.log 2>&1
cd $R; timeout 40 python tools/layer_profile.py --precision f32_bf16x6 --csv $O/layers_x6.csv > $O/layers_x6.log 2>&1
rm -f $O/prof_train/*kernel_trace.csv; ls $O $O/prof_train
