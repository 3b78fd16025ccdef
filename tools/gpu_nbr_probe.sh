#!/bin/bash
# how many sweep launches the device neighbourhood sampler needs on the training graph
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
for n in 8 10 11 12 13 14 16 20 48; do RGCN_NBR_LAUNCHES=$n python tools/nbr_sampler_trace.py 20 2>&1 | tail -2; done
