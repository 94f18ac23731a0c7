for c in C4 C3; do PTK_OPTS=wavefront=1 PTK_STATS_SPP=64 timeout -k 10 150 python tools/stats_probe.py $c 256 | cut -c1-230; done
PTK_OPTS=wavefront=1,wavefront_paths=4194304 timeout -k 10 150 python tools/stats_probe.py C4 256 | cut -c1-100
PTK_OPTS=wavefront=1,wavefront_paths=200000000 timeout -k 10 150 python tools/stats_probe.py C4 256 | cut -c1-100
