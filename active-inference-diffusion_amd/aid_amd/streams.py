"""The three extra streams of the process.

HIP multiplexes the streams of a process onto four hardware queues in order of first use; a fifth stream shares a queue with an earlier one and
two streams on one queue execute in submission order.  For long dependent kernel chains that is a cliff, not a slowdown: a reverse loop whose
stream shares the queue of the stream running the estimator waits behind every one of its kernels (measured: 98 - 108 ms per cfg-2 train_step
instead of 54, depending only on how many streams had been used before; tools/perf/README.md).  So everything in aid_amd that wants a stream
besides the caller's takes it from here: ONE side stream and TWO chain streams per device for the whole process, created once, handed to every
agent - with the caller's stream that is exactly four.  A process that uses further streams of its own is over the limit and should set
`agent.concurrent_belief_loops = False` (belief updates as one merged loop: side + main only).  RCCL's internal stream in data-parallel runs is
such a fifth stream: measured with one rank on real RCCL the chains still paid (50.1 against 56.4 ms merged at the end of round 3), but an N-rank communicator may use
more, so a data-parallel state agent does not assume either way - it times both schedules during its first steps and keeps the faster one
(`DiffusionStateAgent._dp_schedule_vote`)."""
from __future__ import annotations

from typing import Dict, List

import torch

_side: Dict[int, torch.cuda.Stream] = {}
_chains: Dict[int, List[torch.cuda.Stream]] = {}


def _index(device) -> int:
    d = torch.device(device if device is not None else "cuda")
    return d.index if d.index is not None else torch.cuda.current_device()


def side_stream(device=None) -> torch.cuda.Stream:
    """The process's side stream on `device` (high priority: its small dependent kernels get workgroup slots as the other stream's big ones retire)."""
    i = _index(device)
    if i not in _side:
        _side[i] = torch.cuda.Stream(device=torch.device("cuda", i), priority=-1)
    return _side[i]


def chain_streams(device=None) -> List[torch.cuda.Stream]:
    """The process's two extra chain streams on `device`."""
    i = _index(device)
    if i not in _chains:
        import os
        pr = [int(x) for x in os.environ.get("AID_CHAIN_PRIO", "-1,-1").split(",")]          # A/B experiments only
        _chains[i] = [torch.cuda.Stream(device=torch.device("cuda", i), priority=pr[k]) for k in range(2)]
    return _chains[i]
