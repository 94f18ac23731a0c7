for o in "chunk=4" "chunk=8" "chunk=16" "chunk=8,flat_shade_weight=4" "chunk=8,flat_gen_weight=32" "chunk=8,flat_gen_weight=128"; do PTK_OPTS=$o timeout -k 10 100 python tools/stats_probe.py C2 256; done
for o in "chunk=4" "chunk=8" "chunk=16"; do PTK_OPTS=$o timeout -k 10 100 python tools/stats_probe.py C4 256; done
for o in "chunk=2" "chunk=4" "chunk=8"; do PTK_TILE=0,8 PTK_OPTS=$o timeout -k 10 100 python tools/stats_probe.py C2 256; done
for o in "chunk=4" "chunk=8" "chunk=16"; do PTK_TILE=0,8 PTK_OPTS=$o timeout -k 10 100 python tools/stats_probe.py C5 1024; done
for o in "chunk=4" "chunk=8" "chunk=16"; do PTK_OPTS=$o timeout -k 10 100 python tools/stats_probe.py C3 512; done
