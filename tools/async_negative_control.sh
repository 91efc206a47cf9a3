#!/bin/bash
# Run on the GPU box: negative controls of the asynchronous-transport tests (tests/test_gpu_world.py).
# For every event edge between compute and comm streams that the staged multi-rank schedules rest on (offt_host.c, EDGE_*),
# the test build leaves that edge out (OFFT_TEST_DROP_EDGE=<id>) and two ranks as threads run the schedule over the
# asynchronous "slow wire" transport: the result must come out WRONG (or the repeated transform differ) -- the edge is needed
# and the tests would notice its absence.  Last: the same schedules with every edge in place.
# (HIP maps a process's streams onto a few hardware queues -- 4 by default -- and two streams that share one run in order:
#  a dropped edge between them then goes unnoticed, depending on which streams happened to land together.  More queues than
#  streams, so that every dropped edge is a real race)
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-24}
run() { python3 tests/_thread_world.py 2 "$1" /tmp/tw_neg 2>&1 | grep -v "^P1 \|^M1 \|amdgpu.ids" | tail -n 1 | cut -c1-300; }
mkdir -p /tmp/tw_neg
SLAB='[{"N": [256, 256, 256], "params": {"P1": 1, "T1": 32, "T2": 16}, "repeat": 2, "async": 1}]'
SLABINV='[{"N": [256, 256, 256], "params": {"P1": 1, "T1": 32, "T2": 16}, "inv": 1, "repeat": 1, "async": 1}]'
PENCIL='[{"N": [256, 256, 256], "params": {"P1": 2, "T1": 16, "W1": 1, "T2": 16}, "repeat": 2, "async": 1}]'
PENCIL1='[{"N": [256, 256, 256], "params": {"P1": 1, "S": 1, "T1": 16, "W1": 1, "T2": 16}, "repeat": 2, "async": 1}]'
PENCILINV='[{"N": [256, 256, 256], "params": {"P1": 2, "T1": 16, "W1": 1, "T2": 16}, "inv": 1, "repeat": 1, "async": 1}]'
PENCIL1INV='[{"N": [256, 256, 256], "params": {"P1": 1, "S": 1, "T1": 16, "W1": 1, "T2": 16}, "inv": 1, "repeat": 1, "async": 1}]'
# consumer-side edges ("a kernel waits for its exchange"): slow wire (a millisecond of spinning ahead of the copies, "async": 1)
for e in "1 slab:K2-after-exchange $SLAB" "4 pencil:K2-after-exchange1 $PENCIL1" "6 pencil:K3-after-exchange2 $PENCIL" "8 inverse:K1-after-the-exchanges $SLABINV" \
         "12 pencil-inverse:K2-after-exchange2 $PENCILINV" "14 pencil-inverse:K1-after-exchange1 $PENCIL1INV"; do
  set -- $e; id=$1; name=$2; shift 2
  echo "--- edge $id ($name) DROPPED, slow wire:"; OFFT_TEST_DROP_EDGE=$id run "$*"
done
# producer-side edges ("an exchange waits for the kernel that packs its data"): fast wire ("async": 2), slow passes (50 ms of
# spinning ahead of every pass, OFFT_TEST_SLOW_PASS_MS, test build)
fast() { echo "$1" | sed 's/"async": 1/"async": 2/'; }
echo "--- edge 2 (slab:exchange-after-K1) DROPPED, slow passes:"; OFFT_TEST_SLOW_PASS_MS=50 OFFT_TEST_DROP_EDGE=2 run "$(fast "$SLAB")"
echo "--- edge 3 (pencil:exchange1-after-K1) DROPPED, slow passes:"; OFFT_TEST_SLOW_PASS_MS=50 OFFT_TEST_DROP_EDGE=3 run "$(fast "$PENCIL1")"
echo "--- edge 5 (pencil:exchange2-after-K2) DROPPED, slow passes:"; OFFT_TEST_SLOW_PASS_MS=50 OFFT_TEST_DROP_EDGE=5 run "$(fast "$PENCIL")"
echo "--- edge 7 (inverse:exchange-after-its-passes) DROPPED, slow passes:"; OFFT_TEST_SLOW_PASS_MS=50 OFFT_TEST_DROP_EDGE=7 run "$(fast "$SLABINV")"
echo "--- edge 11 (pencil-inverse:exchange2-after-K3) DROPPED, slow passes:"; OFFT_TEST_SLOW_PASS_MS=50 OFFT_TEST_DROP_EDGE=11 run "$(fast "$PENCILINV")"
echo "--- edge 13 (pencil-inverse:exchange1-after-K2) DROPPED, slow passes:"; OFFT_TEST_SLOW_PASS_MS=50 OFFT_TEST_DROP_EDGE=13 run "$(fast "$PENCIL1INV")"
# the direct-store exchange's flag waits, slow passes on the ODD rank only (the even rank runs ahead of it): READY ("the blocks are in my volume") and FREE ("overwrite what you stored")
P2PSLAB='[{"N": [128, 128, 128], "params": {"P1": 1, "T1": 32, "T2": 16}, "p2p": 1, "repeat": 2}]'
P2PPENCIL='[{"N": [128, 128, 128], "params": {"P1": 2, "T1": 16, "W1": 1, "T2": 16}, "p2p": 1, "repeat": 2}]'
for c in "$P2PSLAB" "$P2PPENCIL"; do
  echo "--- edge 9 (direct-store: wait READY) DROPPED, the odd rank slow:"; OFFT_TEST_SLOW_RANKS=odd OFFT_TEST_SLOW_PASS_MS=50 OFFT_TEST_DROP_EDGE=9 run "$c"
  echo "--- edge 10 (direct-store: wait FREE) DROPPED, the odd rank slow:"; OFFT_TEST_SLOW_RANKS=odd OFFT_TEST_SLOW_PASS_MS=50 OFFT_TEST_DROP_EDGE=10 run "$c"
  echo "--- direct-store, every wait in place, the odd rank slow:"; OFFT_TEST_SLOW_RANKS=odd OFFT_TEST_SLOW_PASS_MS=50 run "$c"
done
echo "--- every edge in place, slow wire:"; for c in "$SLAB" "$SLABINV" "$PENCIL" "$PENCIL1" "$PENCILINV" "$PENCIL1INV"; do run "$c"; done
echo "--- every edge in place, slow passes:"; for c in "$SLAB" "$SLABINV" "$PENCIL" "$PENCIL1" "$PENCILINV" "$PENCIL1INV"; do OFFT_TEST_SLOW_PASS_MS=50 run "$(fast "$c")"; done
