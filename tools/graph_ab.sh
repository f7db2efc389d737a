export FLUHIP_AB=1   # the build whose experiment switches are live (flucoma-core_amd/build.py --ab)
for g in 0 10 50 0 10 50; do FLUHIP_GRAPH_ITERS=$g timeout 300 python tools/graph_ab.py 1000 2>&1 | tail -4; done
