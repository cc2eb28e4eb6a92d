#!/bin/bash
# the product CLI end to end on synthetic batches, every compute dtype, captured + device feeder
cd "${GRAFT_REPO_ROOT:-.}"
for dt in f32 bf16s bf16; do
  rm -rf /tmp/cli_$dt
  ( time timeout -k 10 400 python3 train_flownet.py -m /tmp/cli_$dt --flownet_path $PWD/dvs_of_training_framework_amd --optimizer ADAM -bs 8 -mbs 8 --height 256 --width 256 -lr 1e-3 --event-representation-depth 5 --synthetic -ne 300 -d cuda:0 --capture --device-feeder --compute-dtype $dt ) > /tmp/cli_$dt.log 2>&1; rc=$?
  echo "cli $dt rc=$rc $(grep -c . /tmp/cli_$dt.log) lines; $(grep real /tmp/cli_$dt.log)"; tail -3 /tmp/cli_$dt.log | cut -c1-200; ls /tmp/cli_$dt | head -3
  [ $rc -ne 0 ] && exit $rc
done
# gradient accumulation (bs 16 = 2 micro-batches) in f32
rm -rf /tmp/cli_acc
timeout -k 10 400 python3 train_flownet.py -m /tmp/cli_acc --flownet_path $PWD/dvs_of_training_framework_amd --optimizer ADAM -bs 16 -mbs 8 --height 256 --width 256 -lr 1e-3 --event-representation-depth 5 --synthetic -ne 100 -d cuda:0 --capture --device-feeder > /tmp/cli_acc.log 2>&1; echo "cli accumulation rc=$?"; tail -2 /tmp/cli_acc.log | cut -c1-200
