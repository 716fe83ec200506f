#!/bin/bash
# eNB / UE processed rates of the reference's phy_dl_test through the drop-in (for scripts/ab_script.sh)
for i in 1 2 3; do oracle/_ref/hip/phy_dl_test -p 100 -t 1 -m 28 2>&1 | grep -E "eNb:|UE:" | tr '\n' ' '; echo; done
