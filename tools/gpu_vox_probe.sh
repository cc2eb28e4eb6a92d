#!/bin/bash
cd $GRAFT_REPO_ROOT
for e in 4 8 16; do
echo -n "EPT=$e B=64: "; DVSOF_VOX_EPT=$e python3 tools/vox_probe.py 64 5 256 256 65536 2>/dev/null
echo -n "EPT=$e B=4 1M: "; DVSOF_VOX_EPT=$e python3 tools/vox_probe.py 4 12 512 512 1000000 2>/dev/null
echo -n "EPT=$e B=8: "; DVSOF_VOX_EPT=$e python3 tools/vox_probe.py 8 5 256 256 65536 2>/dev/null
done
