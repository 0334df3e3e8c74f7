#!/bin/bash
# Round 6: the kernel choices follow what is in flight, not ugsm_config.slots.  Same-box A/B (tools/ab.py):
#   bash tools/ab_alone.sh out.txt
#   * a 16 MP call alone on the chip: one-slot context against four-slot context (calls one at a time, each waited for: the node's service
#     call and its one-at-a-time topic path), the four-slot context with the choices of a shared chip forced (= rounds 3-5), no side stream,
#     side streams of the slots' own instead of the borrowed ones (UGSM_SIDE_PRIO);
#   * the shape bench.py times (four slots, calls of eight) and four single-pair calls in flight: the library's choice against both forced;
#   * the same at 1080p and for the foveated stack.
out=$1
part=${2:-ACB}   # A: a 16 MP call alone; C: 16 MP calls in flight; B: 1080p and the foveated stack (gpurun calls of <= 20 minutes: A, then CB)
: > $out
export UGSM_DEV=1
ab() { python tools/ab.py "$@" | grep -v "^round" >> $out || exit 1; }
if [[ $part == *A* ]]; then
ab --pairs 32 --rounds 5 "one slot:SLOTS=1" "four slots, one call at a time:SLOTS=4;SERIAL=1" "four slots, one at a time, shared choices forced (rounds 3-5):SLOTS=4;SERIAL=1;UGSM_ALONE=0" "one slot, no side stream:SLOTS=1;UGSM_TWO_STREAMS=0" "one slot, side stream in another priority pool:SLOTS=1;UGSM_SIDE_PRIO=l" "four slots, one at a time, no side stream:SLOTS=4;SERIAL=1;UGSM_TWO_STREAMS=0" "four slots, one at a time, a side stream of its own per slot (before the borrowed ones):SLOTS=4;SERIAL=1;UGSM_SIDE_PRIO=s"
fi
if [[ $part == *C* ]]; then
ab --slots 4 --batch 8 --pairs 64 --rounds 2 "four slots, calls of 8 -- library's choice:" "every call taken to share the chip:UGSM_ALONE=0" "every call taken to be alone:UGSM_ALONE=1"
ab --slots 4 --batch 1 --pairs 64 --rounds 2 "four slots, single-pair calls -- library's choice:" "every call taken to share the chip:UGSM_ALONE=0" "every call taken to be alone:UGSM_ALONE=1" "idle side streams in another priority pool:UGSM_SIDE_PRIO=l" "no side streams:UGSM_TWO_STREAMS=0"
fi
if [[ $part == *B* ]]; then
ab --size 1920 1080 --pairs 128 --rounds 2 "1080p, one slot:SLOTS=1" "1080p, four slots, one call at a time:SLOTS=4;SERIAL=1" "1080p, four slots, one at a time, shared choices forced:SLOTS=4;SERIAL=1;UGSM_ALONE=0"
ab --size 1920 1080 --slots 4 --batch 8 --pairs 256 --rounds 2 "1080p, four slots, calls of 8 -- library's choice:" "every call taken to share the chip:UGSM_ALONE=0" "every call taken to be alone:UGSM_ALONE=1"
ab --fovea 7 --pairs 128 --rounds 2 "foveated 16 MP, one slot:SLOTS=1" "foveated, four slots, one call at a time:SLOTS=4;SERIAL=1" "foveated, four slots, one at a time, shared choices forced:SLOTS=4;SERIAL=1;UGSM_ALONE=0"
ab --fovea 7 --slots 4 --batch 8 --pairs 256 --rounds 2 "foveated, four slots, calls of 8 -- library's choice:" "every call taken to share the chip:UGSM_ALONE=0" "every call taken to be alone:UGSM_ALONE=1"
fi
cat $out
