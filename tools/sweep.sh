for c in C2 C1 C3 C4 C5; do timeout -k 10 100 python tools/stats_probe.py $c 256; done
for o in "chunk=8" "chunk=24" "chunk=32"; do PTK_OPTS=$o timeout -k 10 100 python tools/stats_probe.py C4 256; done
