"""One LUDVM simulation on several GPUs of a node in ONE process, no launcher:

    sim = LUDVM(t0=0, tf=50, dt=1e-3, ..., devices=8)            # or devices=[0, 2, 5]
    sim.Cl, sim.flowfield(...), sim.induced_velocity(...)          # as ever; sim.close() when done

The reference's caller writes `LUDVM(...)` and the run happens inside the constructor (LUDVM.py:231-297); the process-per-GPU
forms (`distributed='rccl'`, ludvm_amd/comm.py; `distributed=True`, ludvm_amd/distributed.py) ask that caller to run the whole
script G times under a launcher.  Here the G "ranks" are G HOST THREADS of this one process:

  * one Engine per device and one replica of the simulation per thread, exactly what a rank process holds: every replica
    keeps the whole wake and runs the whole time loop but evaluates only its tile block of the roll-up's unordered pairs
    (ludvm_amd/distributed.py says what is sharded and what each method exchanges);
  * ONE communicator for the engines, created by one call from the constructing thread -- Engine.comm_init_all =
    ludvm_comm_init_all = ncclCommInitAll: no identifier, no rendezvous file, no environment;
  * the collectives are issued INSIDE the C ABI (between the symmetric kernel and the Euler finisher of
    ludvm_wake_advect* / ludvm_wake_step / ludvm_march_run), each replica's on its own device's stream, from its own thread --
    RCCL's rule for one process is "one thread per device, or grouped calls", and grouping from one thread would mean cutting
    every such entry point in two around its collective.  ctypes releases the interpreter lock for the length of a foreign
    call, so the G threads sit in their C calls concurrently; the host Python between calls (a few hundred microseconds per
    stretch of 512 marched steps) is serialised by the lock and replicated, as it is replicated across processes.

The object the caller holds is a thin front: attributes are replica 0's (every replica holds the same bits: integer sums
commute), method calls run on all replicas together -- they are collective -- and return replica 0's result.  Not an instance of
LUDVM (isinstance says so); everything else the reference's caller touches is there.

What cannot be rehearsed on this build's one-GPU boxes: RCCL refuses two ranks on one device, so the real communicator is
exercised with one device (LUDVM_COMM_FORCE=1 issues its collectives) and the threads + sharding with several engines on ONE card
joined by a test-only in-process all-reduce through ludvm_set_shard (tests/test_gpu_multi.py); on the CPU the same threads
run over tests/fake_engine.py (tests/test_multi_threads.py).
"""
import queue
import threading

import numpy as np


class ThreadTeam:
    """G worker threads, one per rank; run(fn) calls fn(rank) on all of them together and returns the results in rank order.
    An exception on any rank is re-raised here (the lowest rank's), after every rank has finished or failed -- a rank that
    fails inside a collective leaves its peers waiting in RCCL, so a failure is not survivable in general: `broken` is set and
    further run() calls refuse."""

    def __init__(self, world, name="ludvm-rank"):
        self.world = int(world)
        self.broken = None
        self._q = [queue.Queue() for _ in range(self.world)]
        self._done = queue.Queue()
        self._threads = [threading.Thread(target=self._loop, args=(r,), name=f"{name}-{r}", daemon=True) for r in range(self.world)]
        for t in self._threads:
            t.start()

    def _loop(self, rank):
        while True:
            fn = self._q[rank].get()
            if fn is None:
                return
            try:
                self._done.put((rank, fn(rank), None))
            except BaseException as e:      # noqa: BLE001  (handed to the caller of run())
                self._done.put((rank, None, e))

    def run(self, fn, timeout=None):
        if self.broken is not None:
            raise RuntimeError(f"a rank of this device team failed earlier ({self.broken!r}); the team cannot be used any more")
        for q in self._q:
            q.put(fn)
        out, errs = [None] * self.world, {}
        for _ in range(self.world):
            try:
                rank, res, err = self._done.get(timeout=timeout)
            except queue.Empty:
                self.broken = TimeoutError(f"a rank did not return within {timeout} s")
                raise self.broken from None
            out[rank] = res
            if err is not None:
                errs[rank] = err
        if errs:
            r = min(errs)
            self.broken = errs[r]
            raise errs[r]
        return out

    def stop(self):
        for q in self._q:
            q.put(None)
        for t in self._threads:
            t.join(5.0)


def join_with_rccl(engines, min_targets, min_wake, min_pairs):
    """The default way the replicas' engines are joined: the library's own communicator over all of them, one call
    (ncclCommInitAll) -> one LibraryGroup per rank."""
    from .comm import LibraryGroup
    type(engines[0]).comm_init_all(engines, min_wake)
    return [LibraryGroup.joined(e, r, len(engines), min_targets, min_wake, min_pairs) for r, e in enumerate(engines)]


def normalise_devices(devices):
    """devices=8 -> [0 .. 7]; a sequence of ordinals stays; duplicates are refused (one replica per GPU)."""
    devs = list(range(devices)) if isinstance(devices, (int, np.integer)) else [int(d) for d in devices]
    if not devs:
        raise ValueError("devices: at least one device")
    if len(set(devs)) != len(devs):
        raise ValueError(f"devices={devs}: one replica per GPU (RCCL ranks cannot share a device)")
    return devs


class MultiDeviceLUDVM:
    """LUDVM(..., devices=[...]) with more than one device: see the module docstring.  `engine_factory(device) -> engine` and
    `join(engines, min_targets, min_wake, min_pairs) -> [group per rank]` are the two seams the tests use."""

    _OWN = ("_team", "_sims", "_groups", "_engines", "devices", "world", "_closed")

    def __init__(self, args, kwargs, devices, engine_factory=None, join=join_with_rccl, min_targets=None, min_wake=None,
                 min_pairs=None, builder=None):
        from . import comm
        from .ludvm import LUDVM
        if kwargs.get("engine") is not None or kwargs.get("distributed") is not None:
            raise ValueError("devices=[...] creates the engines and their communicator itself: do not pass engine= / distributed=")
        # the reference's caller may pass everything positionally (LUDVM.py:231-236): by name from here on
        import inspect
        names = [n for n in inspect.signature(LUDVM.__init__).parameters][1:]
        if len(args) > len(names):
            raise TypeError("too many positional arguments")
        kwargs = dict(dict(zip(names, args)), **kwargs)
        args = ()
        kwargs = {k: v for k, v in kwargs.items() if k not in ("engine", "distributed", "device", "devices")}
        if engine_factory is None:
            from .engine import Engine
            engine_factory = Engine
        object.__setattr__(self, "devices", [int(d) for d in devices])
        object.__setattr__(self, "world", len(self.devices))
        object.__setattr__(self, "_closed", False)
        object.__setattr__(self, "_team", ThreadTeam(self.world))
        object.__setattr__(self, "_sims", None)
        object.__setattr__(self, "_groups", None)
        try:
            object.__setattr__(self, "_engines", self._team.run(lambda r: engine_factory(self.devices[r])))
            groups = join(self._engines, comm.MIN_TARGETS if min_targets is None else min_targets,
                          comm.MIN_WAKE if min_wake is None else min_wake, comm.MIN_PAIRS if min_pairs is None else min_pairs)
            object.__setattr__(self, "_groups", groups)
            verbose = kwargs.pop("verbose", True)
            if builder is None:
                # (the reference prints its progress once, not G times)
                def builder(r, engine, group):
                    return LUDVM(*args, **kwargs, verbose=verbose and r == 0, engine=engine, distributed=group)
            # builder(rank, engine, group) -> that rank's replica (LUDVM.resume(..., devices=...) passes its own)
            sims = self._team.run(lambda r: builder(r, self._engines[r], groups[r]))
            object.__setattr__(self, "_sims", sims)
        except BaseException:
            self._shutdown(wait_for_peers=False)
            raise

    # ---- the front: replica 0's attributes, collective method calls ------------------------------------------------------------
    def __getattr__(self, name):
        sims = object.__getattribute__(self, "_sims")
        if sims is None:
            raise AttributeError(name)
        attr = getattr(sims[0], name)
        if not callable(attr) or isinstance(attr, type):
            return attr
        team = object.__getattribute__(self, "_team")

        def collective(*a, **k):
            if object.__getattribute__(self, "_closed"):      # after close(): replica 0 alone, on its own (unsharded) engine
                sims[0]._shard = None
                return getattr(sims[0], name)(*a, **k)
            return team.run(lambda r: getattr(sims[r], name)(*a, **k))[0]
        collective.__name__ = name
        collective.__doc__ = getattr(attr, "__doc__", None)
        return collective

    def __setattr__(self, name, value):
        if name in MultiDeviceLUDVM._OWN:
            return object.__setattr__(self, name, value)
        for s in self._sims:                  # a parameter the caller changes (verbose, v_core ...) changes on every replica
            setattr(s, name, value)

    def replicas(self):
        """The G replicas (tests: they hold the same bits)."""
        return list(self._sims)

    # ---- teardown --------------------------------------------------------------------------------------------------------------
    def _shutdown(self, wait_for_peers=True):
        if self._closed:
            return
        object.__setattr__(self, "_closed", True)
        team, groups = self._team, self._groups
        alive = all(t.is_alive() for t in team._threads)      # (a finalizer at interpreter exit finds the daemon threads gone)
        if groups is not None and team.broken is None and wait_for_peers and alive:
            try:
                team.run(lambda r: groups[r].close(), timeout=60.0)      # (ncclCommDestroy: every rank, together)
            except BaseException:       # noqa: BLE001
                pass
        if alive:
            team.stop()

    def close(self):
        """Leave the communicator and end the rank threads; the results stay readable."""
        self._shutdown()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self._shutdown()
        except BaseException:           # noqa: BLE001
            pass
