for c in C2 C3 C4 C5; do PTK_STATS_SPP=64 timeout -k 10 100 python tools/stats_probe.py $c 256 | cut -c1-200; done
