for r in 1 2; do for v in base ntl nts ntls; do echo "== $v"; OFFT_AMD_LIB=$PWD/build/dev/$v/liboffthip.so timeout -k 10 120 python tools/dev_perf.py 1024 zyx 1 2>&1 | grep "N="; done; done
