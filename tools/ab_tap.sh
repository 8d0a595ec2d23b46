L="spade_shared_256 up2_conv0 mid_conv_16 flow_res_32 spade_shared_64 g_mid_16_1024_128 spade_gamma_256_c64"
python tools/bench_conv.py $L 2>/dev/null | grep -E '"fwd"|"dgrad"' > gpurun_out/ab_tap.log
AG2V_NO_TAP_MAJOR=1 python tools/bench_conv.py $L 2>/dev/null | grep -E '"fwd"|"dgrad"' > gpurun_out/ab_notap.log
paste gpurun_out/ab_tap.log gpurun_out/ab_notap.log | awk '{print $2,$4,$8,"|",$16}'
for i in 1 2; do
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-kernel-timing 2>/dev/null | tail -1 | cut -c95-200
AG2V_NO_TAP_MAJOR=1 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-kernel-timing 2>/dev/null | tail -1 | cut -c95-200
done
