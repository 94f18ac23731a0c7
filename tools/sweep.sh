for c in C2 C4 C3; do PTK_STATS_SPP=64 timeout -k 10 100 python tools/stats_probe.py $c 256 | cut -c1-100; done
