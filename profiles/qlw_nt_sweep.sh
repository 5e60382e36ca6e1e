for nt in 64 128 256; do echo "n=32 NT=$nt"; TBK_QLW_NT=$nt QLW_SIZES=32 QLW_TIMING_ONLY=1 python3 profiles/qlw_probe.py child; done
for nt in 128 256 512; do echo "n=64 NT=$nt"; TBK_QLW_NT=$nt QLW_SIZES=64 QLW_TIMING_ONLY=1 python3 profiles/qlw_probe.py child; done
for nt in 128 256 512; do echo "n=48 NT=$nt"; TBK_QLW_NT=$nt QLW_SIZES=48 QLW_TIMING_ONLY=1 python3 profiles/qlw_probe.py child; done
for nt in 64 128 256; do echo "n=20 NT=$nt"; TBK_QLW_NT=$nt QLW_SIZES=20 QLW_TIMING_ONLY=1 python3 profiles/qlw_probe.py child; done
