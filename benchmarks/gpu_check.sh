bash benchmarks/profile_round2.sh > gpurun_out/profile_round2.out 2>&1; tail -5 gpurun_out/profile_round2.out | cut -c1-300
