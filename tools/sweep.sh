for c in C2 C1 C3 C4 C5; do timeout -k 10 100 python tools/stats_probe.py $c 256; done
for r in 0 1 2 3 4 5 6 7; do PTK_TILE=$r,8 timeout -k 10 100 python tools/stats_probe.py C2 256; done
for r in 0 1 2 3; do PTK_TILE=$r,4 timeout -k 10 100 python tools/stats_probe.py C2 256; done
for r in 0 1; do PTK_TILE=$r,2 timeout -k 10 100 python tools/stats_probe.py C2 256; done
