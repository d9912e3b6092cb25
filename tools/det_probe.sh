run3() { fg=$1; bg=$2; (DET_MODES=$bg python tools/determinism_check.py 96 12 A > /tmp/a.txt 2>&1 &) ; (DET_MODES=$bg python tools/determinism_check.py 96 12 B > /tmp/b.txt 2>&1 &); DET_MODES=$fg python tools/determinism_check.py 96 12 C 2>&1 | tail -3 | cut -c1-100; sleep 5; }
for i in 1 2 3; do echo "fg mode 0, bg mode 3 (conv_nin_h only)"; run3 0 3; done
for i in 1 2 3; do echo "fg mode 3, bg mode 2"; run3 3 2; done
