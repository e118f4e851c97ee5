#!/bin/bash
# Round 6, VERDICT r5 item 3(a): 128x64 / 64x128 / 128x128 tiles COMBINED with the exact split-K (groups re-planned per tile shape) on the dominant f32 shapes, stand-alone.
# Stop rule: best-of < 0.62 of the MFMA peak on the s2 3x3 shape -> written down, no further f32 k-loop work.
TAG=${1:-r09b}
O=gpurun_out/$TAG
mkdir -p $O
V27="27:2:9:3"
sweep() { # layer, plans
  timeout 300 python tools/layer_probe.py --layers $1 --variants $2 --reps 20 >> $O/bigtiles.txt 2>&1
}
echo "# s2 3x3 (M 256, N 6272, K 2304 = 9 depth blocks): committed plan 27:2:9:3" > $O/bigtiles.txt
sweep s2b1c2 "27:2:9:3,3:2:9:3,0:2:9:0,0:2:9:3,0:2:3:0,0:2:3:3,0:1:9:0,1:2:9:0,1:2:9:3,1:2:3:0,1:2:3:3,1:2:4:3,2:2:9:0,2:2:9:3,2:2:3:0,2:2:3:3,2:2:4:3,12:2:9:3,13:2:9:3,13:2:3:3,14:2:9:3,14:2:3:3,16:2:9:3,17:2:3:3,18:2:3:3,1:1:3:0,2:1:3:0"
echo "# s1 3x3 (M 128, N 25088, K 1152 = 5 depth blocks): committed 3:1:5:0" >> $O/bigtiles.txt
sweep s1b1c2 "3:1:5:0,0:2:5:0,0:2:5:3,0:2:2:3,0:1:5:0,1:2:5:3,1:2:2:3,1:1:5:0,2:2:5:3,2:2:2:3,2:1:5:0,1:0:1:0,2:0:1:0,0:0:1:0"
echo "# s3 3x3 (M 512, N 1568, K 4608 = 18 depth blocks): committed 27:2:6:3" >> $O/bigtiles.txt
sweep s3b1c2 "27:2:6:3,0:2:18:3,0:2:9:3,0:2:6:3,1:2:9:3,1:2:6:3,1:2:18:3,2:2:9:3,2:2:6:3,2:2:18:3"
echo "# s2 1x1 K 1024 (M 256, N 6272, 4 depth blocks): committed 3:1:4:0" >> $O/bigtiles.txt
sweep s2b1c1 "3:1:4:0,0:2:4:3,0:2:4:0,1:2:4:3,1:2:2:3,2:2:4:3,2:2:2:3,1:1:4:0,2:1:4:0"
echo "# s2 1x1 expand (M 1024, N 6272, K 256): committed 27:0:1:0" >> $O/bigtiles.txt
sweep s2b1c3 "27:0:1:0,0:0:1:0,1:0:1:0,2:0:1:0,0:0:1:1,1:0:1:1,2:0:1:1,12:0:1:0,13:0:1:0,14:0:1:0"
cat $O/bigtiles.txt
