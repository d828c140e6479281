"""Host side of the cross-chain step-size statistics and of the reference's step-size adaptation
(barlib/src/hmc/leapfrog/time_step.cpp), all without a GPU:

  * bchmc_eps_exchange through the C entry point with a stub transport: record packing, queueing of more than
    BCHMC_EPS_BATCH records, validation of what the transport returns;
  * the compiled C++ adaptation (barcode_amd/shim/eps_adapt.cc) against the numpy restatement (barcode_amd/time_step.py)
    on random histories, and both against the hand-worked cases of tests/test_time_step.py;
  * the C++ HamiltonianMC loop's bookkeeping on a scripted engine: update_eps_fac before every trajectory (HMC.cc:453),
    rejections (500-501), the tables (506-507), scheme 3's halving until the first acceptance (time_step.cpp:137-149);
  * the torch.distributed transport with gloo: world sizes 2 and 3, UNEQUAL numbers of attempts per rank and sample.
"""
import os
import socket
import struct
import subprocess
import sys

import numpy as np
import pytest

from barcode_amd import engine as eng
from barcode_amd import time_step as ts
from barcode_amd.chains import EpsRing
from barcode_amd.params import HamilParams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------------------------------------------------
# bchmc_eps_exchange with a stub transport
# ---------------------------------------------------------------------------------------------------------------
def _packet(records):
    """Wire format of one rank's contribution (eps_comm.hip Packet): int32 n, int32 0, 32 x {double, int32, int32}."""
    assert len(records) <= eng.EPS_BATCH
    b = struct.pack("<ii", len(records), 0)
    for e, a, n in records:
        b += struct.pack("<dii", e, int(a), n)
    return b + b"\0" * (eng.PACKET_BYTES - len(b))


def test_exchange_packs_records_and_returns_them_in_rank_order():
    others = {0: [(0.5, True, 3)], 2: [(0.25, False, 7), (0.125, True, 1)]}
    seen = []

    def allgather(send):
        seen.append(send)
        return _packet(others[0]) + send + _packet(others[2])

    c = eng.Comm(rank=1, world=3, allgather=allgather)
    got = c.exchange([(0.75, False, 4), (1.5, True, 8)])
    assert seen[0] == _packet([(0.75, False, 4), (1.5, True, 8)])          # packing
    assert got == [(0, 0.5, True, 3), (1, 0.75, False, 4), (1, 1.5, True, 8), (2, 0.25, False, 7), (2, 0.125, True, 1)]
    assert c.pending() == 0
    assert c.exchange([]) == [(0, 0.5, True, 3), (2, 0.25, False, 7), (2, 0.125, True, 1)]  # empty contribution
    c.close()


def test_exchange_queues_what_does_not_fit_one_batch():
    sent = []

    def allgather(send):
        sent.append(struct.unpack_from("<i", send)[0])
        return send + _packet([])

    c = eng.Comm(rank=0, world=2, allgather=allgather)
    recs = [(0.01 * i, i % 2 == 0, i) for i in range(eng.EPS_BATCH + 5)]
    first = c.exchange(recs)
    assert sent == [eng.EPS_BATCH] and c.pending() == 5
    assert [r[3] for r in first] == list(range(eng.EPS_BATCH))             # oldest first
    second = c.exchange([(9.0, True, 99)])
    assert sent[-1] == 6 and c.pending() == 0
    assert [r[3] for r in second] == list(range(eng.EPS_BATCH, eng.EPS_BATCH + 5)) + [99]
    c.close()


def test_exchange_rejects_a_transport_that_scribbles():
    bad_count = struct.pack("<ii", 1000, 0) + b"\0" * (eng.PACKET_BYTES - 8)
    c = eng.Comm(rank=0, world=2, allgather=lambda send: send + bad_count)
    with pytest.raises(eng.BchmcError, match="malformed"):
        c.exchange([(1.0, True, 1)])
    assert c.pending() == 0        # a failed call keeps none of its records queued: the caller retries with them
    c.close()
    c2 = eng.Comm(rank=0, world=2, allgather=lambda send: b"short")
    with pytest.raises(eng.BchmcError):
        c2.exchange([(1.0, True, 1)])
    c2.close()


def test_single_rank_communicator_needs_no_transport():
    c = eng.Comm(rank=0, world=1)
    assert c.exchange([(0.5, True, 2)]) == [(0, 0.5, True, 2)]
    c.close()


# ---------------------------------------------------------------------------------------------------------------
# C++ step-size adaptation == numpy restatement
# ---------------------------------------------------------------------------------------------------------------
def _shim(eps_fac=1.0, cfg=None):
    from barcode_amd.shim import ShimHamil
    hd = ShimHamil(HamilParams(Nx=8, L=25.0), N_eps_fac=8.0, eps_fac=eps_fac)
    if cfg is not None:
        hd.eps_attach(cfg)
    return hd


@pytest.mark.parametrize("update_type", [1, 2, 3])
def test_cpp_update_eps_fac_equals_the_numpy_restatement_on_random_histories(update_type):
    rng = np.random.default_rng(100 + update_type)
    for trial in range(30):
        n_a = int(rng.integers(4, 24))
        cfg = ts.EpsConfig(eps_fac_update_type=update_type, N_a_eps_update=n_a, acc_min=0.6, acc_max=0.7,
                           eps_down_smooth=int(rng.integers(0, 4)), eps_up_fac=float(rng.uniform(0.8, 1.2)),
                           eps_fac_target=0.3, eps_fac_power=float(rng.choice([0.0, 1.0, 2.0])),
                           s_eps_total=int(rng.integers(2, 9)))
        hd = _shim(eps_fac=1.0, cfg=cfg)
        ring, eps_fac = EpsRing(n_a), 1.0
        p_acc = float(rng.choice([0.1, 0.65, 0.95]))
        rejections = 0
        for attempt in range(4 * n_a):
            iG = 1 if attempt < 3 else 2
            hd.numerical.iGibbs, hd.numerical.rejections = iG, rejections
            eps_fac = ts.update_eps_fac(eps_fac, ring, cfg, iGibbs=iG, rejections=rejections)
            hd.update_eps_fac()
            assert hd.numerical.eps_fac == pytest.approx(eps_fac, rel=1e-15), (trial, attempt)
            eps = eps_fac * float(rng.random())                         # distinct epsilons: no sort ties
            acc = bool(rng.random() < p_acc * (1.2 - eps / max(eps_fac, 1e-300)))
            rejections = 0 if acc else rejections + 1
            ring.record(acc, eps)
            hd.numerical.accepted, hd.numerical.epsilon = acc, eps
            hd.update_epsilon_acc_rate_tables()
            # now and then a burst of pooled records from "other chains"
            if rng.random() < 0.2:
                for _ in range(int(rng.integers(1, 2 * n_a))):
                    e2, a2 = float(eps_fac * rng.random()), bool(rng.random() < p_acc)
                    ring.record(a2, e2)
                    hd.eps_append(a2, e2)
            assert hd.eps_records() == ring.count_attempts
            assert hd.eps_acceptance_rate() == ring.acceptance_rate()
        hd.close()


def test_cpp_adaptation_hand_worked_cases():
    """The cases of tests/test_time_step.py through the compiled code."""
    cfg = ts.EpsConfig(eps_fac_update_type=2, N_a_eps_update=10, eps_down_smooth=0)
    hd = _shim(0.5, cfg)
    for acc, eps in [(1, .01), (1, .02), (1, .03), (0, .04), (0, .05), (0, .06), (0, .07), (0, .08), (0, .09), (0, .10)]:
        hd.eps_append(acc, eps)
    msg = hd.update_eps_fac()
    assert np.isclose(hd.numerical.eps_fac, 0.05) and "downwards" in msg   # time_step.cpp:43,86
    hd.close()
    cfg = ts.EpsConfig(eps_fac_update_type=2, N_a_eps_update=4, eps_up_fac=1.0)
    hd = _shim(2.0, cfg)
    for eps in (.1, .2, .3, .4):
        hd.eps_append(True, eps)
    assert "upwards" in hd.update_eps_fac() and np.isclose(hd.numerical.eps_fac, 2.0 / 0.65)
    assert hd.update_eps_fac() == "" and np.isclose(hd.numerical.eps_fac, 2.0 / 0.65)   # fires once per multiple
    hd.close()
    cfg = ts.EpsConfig(eps_fac_update_type=2, N_a_eps_update=4, eps_down_smooth=0)
    hd = _shim(1.0, cfg)
    for _ in range(4):
        hd.eps_append(False, 0.0)
    from barcode_amd.shim import ShimError
    with pytest.raises(ShimError, match="epsilon became zero"):
        hd.update_eps_fac()
    hd.close()


# ---------------------------------------------------------------------------------------------------------------
# HamiltonianMC's bookkeeping on a scripted engine
# ---------------------------------------------------------------------------------------------------------------
def test_scheme3_halves_eps_fac_on_every_rejection_of_the_first_sample():
    """a bad initial eps_fac must not run to itmax.  Script: dH = 50 (p_acc ~ 2e-22: always rejected) for
    the first 5 attempts, then dH = -1 (accepted without a draw).  uniform() feeds Neps, epsilon and the acceptance
    draw in the reference's order (HMC.cc:260-261, 478-480)."""
    cfg = ts.EpsConfig(eps_fac_update_type=3, N_a_eps_update=100)
    hd = _shim(eps_fac=1.6, cfg=cfg)
    hd.numerical.iGibbs = 1
    draws = []

    def uniform():
        draws.append(1)
        return 0.5

    n, log = hd.HamiltonianMC_scripted([50.0] * 5 + [-1.0], uniform, itmax=2000)
    assert n == 6 and [r["accepted"] for r in log] == [False] * 5 + [True]
    # attempt k (k >= 1) runs after k rejections: eps_fac halves before each of them, epsilon = eps_fac * 0.5
    assert [r["epsilon"] for r in log] == [1.6 * 0.5 ** k * 0.5 for k in range(6)]
    assert [r["Neps"] for r in log] == [5] * 6
    assert hd.numerical.rejections == 5 and hd.numerical.accepted and hd.count_attempts.value == 6
    assert hd.numerical.eps_fac == 1.6 / 32
    assert hd.eps_records() == 6 and hd.eps_acceptance_rate() == 1 / 100
    assert len(draws) == 6 * 2 + 5          # the accepted attempt (dH < 0) consumes no acceptance draw
    # second sample of the chain: no more halving, the tables decide (time_step.cpp:141-148)
    hd.numerical.iGibbs, hd.numerical.rejections = 2, 0
    n2, log2 = hd.HamiltonianMC_scripted([50.0, 50.0, 0.0], uniform)
    assert n2 == 3 and [r["epsilon"] for r in log2] == [1.6 / 32 * 0.5] * 3
    hd.close()


def test_loop_stops_at_itmax_and_tolerates_a_short_or_absent_log():
    hd = _shim(eps_fac=1.0, cfg=ts.EpsConfig(eps_fac_update_type=0))
    n, log = hd.HamiltonianMC_scripted([50.0] * 10, lambda: 0.5, itmax=4, log_cap=2)
    assert n == 4 and len(log) == 2 and hd.numerical.rejections == 4
    n, log = hd.HamiltonianMC_scripted([50.0] * 10, lambda: 0.5, itmax=3, log_cap=0)   # log == NULL
    assert n == 3 and log == []
    hd.close()


def test_pooled_records_enter_the_tables_once_per_sample():
    """hd->comm set: ONE exchange after the loop; the other chain's records go into the tables, the own ones are
    not entered twice."""
    calls = []

    def allgather(send):
        calls.append(struct.unpack_from("<i", send)[0])
        return send + _packet([(0.11, True, 3), (0.12, True, 4), (0.13, False, 5)])

    comm = eng.Comm(rank=0, world=2, allgather=allgather)
    hd = _shim(eps_fac=1.0, cfg=ts.EpsConfig(eps_fac_update_type=2, N_a_eps_update=8))
    hd.comm_attach(comm.h, 0)
    n, _ = hd.HamiltonianMC_scripted([50.0, 50.0, -1.0], lambda: 0.5)
    assert n == 3 and calls == [3]                    # three own records in one exchange
    assert hd.eps_records() == 3 + 3
    n, _ = hd.HamiltonianMC_scripted([-1.0], lambda: 0.5)
    assert calls == [3, 1] and hd.eps_records() == 6 + 1 + 3
    hd.comm_attach(None, 0)
    hd.close()
    comm.close()


# ---------------------------------------------------------------------------------------------------------------
# torch.distributed transport (gloo): unequal attempts per rank
# ---------------------------------------------------------------------------------------------------------------
_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from barcode_amd.chains import ChainGroup, EpsRing
rank, world = int(sys.argv[1]), int(sys.argv[2])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=rank, world_size=world)
g = ChainGroup(pool=True)
ring = EpsRing(8)
assert g.chain_seed(1004) == 1004 + rank
total = 0
for sample in range(4):
    # UNEQUAL attempt counts: rank r needs 1 + (r + sample) %% 3 attempts for this sample, one of them 40 (> one batch)
    n_att = 40 if (rank == 1 and sample == 2) else 1 + (rank + sample) %% 3
    mine = [(0.1 * (rank + 1) + 0.001 * a, a == n_att - 1, 5 + rank) for a in range(n_att)]
    for e, acc, _ in mine:
        ring.record(acc, e)                      # own attempts enter the ring as they happen
    got = g.pool_into(ring, mine)                # ONE collective per sample on every rank
    ranks = sorted(set(r for r, *_ in got))
    assert ranks == list(range(world)), got
    for rk, e, acc, neps in got:
        assert neps == 5 + rk and abs(e - 0.1 * (rk + 1)) < 0.05
    total += len(got)
    # the 40-attempt sample does not fit one batch: its tail waits for the next exchange
    assert g.comm.pending() == (8 if (rank == 1 and sample == 2) else 0), (sample, g.comm.pending())
while True:                                      # drain: still one collective per call on every rank
    got = g.pool_into(ring, [])
    total += len(got)
    if not got:
        break
expected = sum(40 if (r == 1 and s == 2) else 1 + (r + s) %% 3 for r in range(world) for s in range(4))
assert total == expected, (total, expected)
assert ring.count_attempts == expected           # own (recorded directly) + pooled: every record exactly once
assert g.broadcast_eps_fac(0.25 if rank == 0 else 9.0) == 0.25
solo = ChainGroup(pool=False)
r2 = EpsRing(8)
assert solo.pool_into(r2, [(0.5, True, 3)]) == [(rank, 0.5, True, 3)] and r2.count_attempts == 0
g.close()
dist.destroy_process_group()
print("ok", rank)
'''


@pytest.mark.parametrize("world", [2, 3])
def test_eps_stats_exchange_gloo_unequal_attempts(tmp_path, world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % dict(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(world)]
    try:
        outs = [p.communicate(timeout=240)[0].decode() for p in procs]   # a mismatched collective would hang: bounded
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "ok" in o


# ---- file bootstrap of the RCCL unique id (bchmc_shim::bootstrap_exchange_id; host only) ---------------------------
def _bootstrap_ranks(path, world, uid, delays, timeout_s=20.0):
    """Run the protocol for all ranks in threads (ctypes releases the GIL in the foreign call)."""
    import threading
    import time
    from barcode_amd import shim
    out, errs = [None] * world, [None] * world

    def run(r):
        time.sleep(delays[r])
        try:
            out[r] = shim.bootstrap_exchange_id(path, r, world, uid if r == 0 else None, timeout_s)
        except shim.ShimError as e:
            errs[r] = str(e)

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return out, errs


def test_bootstrap_file_protocol_delivers_rank0_id(tmp_path):
    from barcode_amd import shim
    path = str(tmp_path / "chain.id")
    uid = bytes(range(128))
    out, errs = _bootstrap_ranks(path, 3, uid, [0.0, 0.05, 0.1])
    assert errs == [None] * 3 and out == [uid] * 3
    for r in range(3):
        shim.bootstrap_cleanup(path, r)
    assert os.listdir(str(tmp_path)) == []       # every rank removed its own files


def test_bootstrap_ignores_the_files_of_an_earlier_run(tmp_path):
    """the second run at a fixed path must not pick up the first run's id (ncclCommInitRank would block on
    a dead address).  Stale files of every kind are in place: a raw 128-byte id (the r02 format), a well-formed id
    file and want / ack files with the nonces of dead processes -- in both start orders."""
    import struct
    path = str(tmp_path / "chain.id")
    magic = 0x62636d63626f6f74
    old_id, new_id = bytes([7] * 128), bytes(range(100, 228))

    def plant():
        open(path, "wb").write(struct.pack("<3Q", magic, 3, 111) + old_id + struct.pack("<3Q", 0, 222, 333))
        open(path + ".want.1", "wb").write(struct.pack("<2Q", magic, 222))
        open(path + ".want.2", "wb").write(struct.pack("<2Q", magic, 333))
        open(path + ".ack.1", "wb").write(struct.pack("<3Q", magic, 222, 111))
        open(path + ".ack.2", "wb").write(struct.pack("<3Q", magic, 333, 111))

    plant()
    out, errs = _bootstrap_ranks(path, 3, new_id, [0.0, 0.3, 0.5])     # rank 0 first: it sees the stale want files
    assert errs == [None] * 3 and out == [new_id] * 3
    plant()
    out, errs = _bootstrap_ranks(path, 3, new_id, [0.4, 0.0, 0.1])     # the others first: they see the stale id file
    assert errs == [None] * 3 and out == [new_id] * 3
    open(path, "wb").write(old_id)                                       # the r02 format: 128 raw bytes
    out, errs = _bootstrap_ranks(path, 2, new_id, [0.3, 0.0])
    assert errs == [None] * 2 and out == [new_id] * 2


def test_bootstrap_times_out_instead_of_hanging(tmp_path):
    from barcode_amd import shim
    path = str(tmp_path / "chain.id")
    with pytest.raises(shim.ShimError, match="timed out"):
        shim.bootstrap_exchange_id(path, 1, 2, None, timeout_s=0.3)      # no rank 0
    with pytest.raises(shim.ShimError, match="timed out"):
        shim.bootstrap_exchange_id(path, 0, 2, bytes(128), timeout_s=0.3)  # rank 1's want file is there, rank 1 is not
    with pytest.raises(shim.ShimError, match="bad argument"):
        shim.bootstrap_exchange_id(path, 2, 2, None, timeout_s=0.1)


def test_exchange_failure_keeps_nothing_queued_and_checks_capacity_first():
    """a failing bchmc_eps_exchange must not leave this call's records queued (a retry would send them
    twice), and the capacity check comes before anything is sent."""
    import ctypes as C
    calls = []

    def broken(payload):
        calls.append(len(payload))
        raise RuntimeError("transport down")

    c = eng.Comm(rank=0, world=2, allgather=broken)
    assert c.info() == dict(world=2, rank=0, transport="custom")
    with pytest.raises(eng.BchmcError):
        c.exchange([(0.5, True, 3)])
    assert calls == [eng.PACKET_BYTES] and c.pending() == 0
    # too small an output array: refused before the transport is touched
    out, who, got = (eng.EpsRecord * 8)(), (C.c_int * 8)(), C.c_int(0)
    mine = (eng.EpsRecord * 1)()
    rc = c.lib.bchmc_eps_exchange(c.h, mine, 1, out, who, 8, C.byref(got))
    assert rc == 1 and len(calls) == 1 and c.pending() == 0 and b"capacity" in c.lib.bchmc_comm_last_error(c.h)
    c.close()
    solo = eng.Comm(rank=0, world=1)
    assert solo.info()["transport"] == "none" and solo.exchange([(0.1, False, 2)]) == [(0, 0.1, False, 2)]
    solo.close()
