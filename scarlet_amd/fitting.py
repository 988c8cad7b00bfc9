"""``fit_blends``: many ``Blend`` objects fitted in one device batch per group of equal frame and
kernel shapes (reference: the per-blend loop of scarlet/testing/api.py:216-224, every blend through
``Blend.fit``, blend.py:85-198).  The batch stays on the device for the whole call: resize test
and resize on the device, every blend at its own iteration counter and pausing at its own
hooks (``_fit_group_resident``); ``Blend.fit`` takes the same path for one blend."""

import logging
import os

import numpy as np

from .batch import BlendBatch
from .blend import Blend, _adaprox_options, _flatten
from .component import CombinedComponent, FactorizedComponent
from .model import Model, UpdateException
from .morphology import ImageMorphology, Morphology, _edge_pull
from .parameter import Parameter, STD_FROM_V
from .renderer import ConvolutionRenderer, NullRenderer, ResolutionRenderer

logger = logging.getLogger("scarlet_amd.blend")


def _export_state(blend):
    """What a fit changes on a Blend, as plain picklable data: the loss history and, per
    factorized component, the box of the morphology and both children's Parameters
    (values, m / v / vhat / std, halved steps after a resize)."""
    comps = _flatten(blend.sources)
    return dict(loss=list(blend.loss),
                comps=[(c.children[1].bbox.origin, c.children[1].bbox.shape,
                        c.children[0]._parameters, c.children[1]._parameters) for c in comps])


def _refresh_boxes(node):
    """Boxes of the containers after their morphologies changed (component.py:172-185,
    280-290: only a container below which a box changed gets a new one)."""
    if isinstance(node, FactorizedComponent):
        box = node._joint_box(*node.children)
        changed = box != node.bbox
        if changed:
            node.bbox = box
        return changed
    changed = [_refresh_boxes(c) for c in node.children]
    if any(changed) and isinstance(node, CombinedComponent):
        node.bbox = node._union_box()
    return any(changed)


def _import_state(blend, state):
    """Apply ``_export_state`` of another process' copy of ``blend``."""
    blend.loss[:] = state["loss"]
    for comp, (origin, shape, p_spec, p_morph) in zip(_flatten(blend.sources), state["comps"]):
        spectrum, morphology = comp.children
        morphology.bbox.origin, morphology.bbox.shape = tuple(origin), tuple(shape)
        for mine, theirs in zip(spectrum._parameters, p_spec):
            mine[...] = theirs
            mine.__dict__.update(theirs.__dict__)
        # a resized image is a new Parameter (morphology.py:155-163, 180-193); a parameter
        # the morphology also holds by name (``shift``) keeps its identity
        new = []
        for mine, theirs in zip(morphology._parameters, p_morph):
            if mine.shape == theirs.shape:
                mine[...] = theirs
                mine.__dict__.update(theirs.__dict__)
                new.append(mine)
            else:
                new.append(theirs)
        morphology._parameters = tuple(new)
    for src in blend.sources:
        _refresh_boxes(src)


def fit_blends(blends, max_iter=200, e_rel=1e-3, min_iter=1, devices=None, **alg_kwargs):
    """Fit many independent ``Blend`` objects together.

    Equivalent to ``[b.fit(max_iter, e_rel, min_iter, **alg_kwargs) for b in blends]``
    -- same per-blend iteration counts, losses, parameter and optimizer-state side
    effects -- but blends that share the frame shape and the difference-kernel stamp run
    in one device batch, so every kernel launch works on all of them (the batched path
    the benchmark measures, behind the reference's per-blend API).  The box-resizing hook
    and the restart it triggers (blend.py:196-198, 284-292) stay per blend: after every
    round the blends are regrouped by their own iteration counter.

    ``devices``: where the blends run (the reference's unit is the per-blend
    loop of ``testing/api.py:216-224``).  ``None`` / an int: one GPU.  A list of GPU
    indices: contiguous shards ``dist.shard_range(len(blends), i, len(devices))``, one
    host thread per GPU.  ``"ranks"``: one process per GPU under ``torch.distributed``
    (every rank holds the same list of blends): rank r fits its shard on GPU LOCAL_RANK
    and the fitted state of every blend is all-gathered, so all ranks return the same
    results and hold the same parameters.  Results do not depend on the partition.

    Returns the list of ``(n_iter, logL)`` tuples; a blend whose parameters turned
    non-finite gets ``(n_iter, nan)`` (and keeps the state of its last iteration), the
    others continue; ``fit_blends.errors`` lists ``(index, ArithmeticError)`` of the last call.
    """
    blends = list(blends)
    kw = dict(max_iter=max_iter, e_rel=e_rel, min_iter=min_iter, **alg_kwargs)
    fit_blends.errors = []
    if isinstance(devices, str):
        if devices != "ranks":
            raise ValueError("devices must be None, an int, a list of GPU indices or 'ranks'")
        from . import dist as sdist

        rank, local_rank, world = sdist.env_rank()
        if not sdist._active():
            if world > 1:
                raise RuntimeError("fit_blends(devices='ranks') under WORLD_SIZE={} needs an "
                                   "initialised process group (scarlet_amd.dist."
                                   "init_process_group)".format(world))
            rank, world = 0, 1
        lo, hi = sdist.shard_range(len(blends), rank, world)
        # a rank that fails must still take part in the collective, or the others hang in it:
        # its exception travels as text and every rank raises together
        try:
            mine, errs = _fit_blends_on(blends[lo:hi], local_rank, **kw)
            part = dict(lo=lo, results=mine, errors=[(lo + i, str(e)) for i, e in errs],
                        states=[_export_state(b) for b in blends[lo:hi]], failed=None)
            import pickle

            pickle.dumps(part)
        except Exception as e:  # noqa: BLE001 -- re-raised on every rank below
            part = dict(lo=lo, results=[], errors=[], states=[],
                        failed="rank {}: {}: {}".format(rank, type(e).__name__, e))
        parts = sdist.gather_objects(part)
        failed = [p["failed"] for p in parts if p["failed"]]
        if failed:
            raise RuntimeError("fit_blends(devices='ranks') failed: " + "; ".join(failed))
        out = []
        for part in parts:
            out.extend(part["results"])
            fit_blends.errors.extend((i, ArithmeticError(msg)) for i, msg in part["errors"])
            if part["lo"] != lo:
                for blend, state in zip(blends[part["lo"]:], part["states"]):
                    _import_state(blend, state)
                    for p in blend.parameters:
                        if p.v is not None:
                            p.std = STD_FROM_V
        return out
    if devices is None or np.isscalar(devices) or len(devices) == 1:
        device = 0 if devices is None else int(devices if np.isscalar(devices) else devices[0])
        # A thousand blends are a few million live Python objects, none of them garbage; the
        # cyclic collector would walk them again and again while the loop below allocates
        # its temporaries (measured: a quarter of the call).  Reference counting still frees
        # everything the loop drops.
        # gc.freeze() takes what exists now out of the collector's generations for the duration
        # of the call; the collector itself stays on (other threads, callbacks).  gc.unfreeze()
        # empties the WHOLE permanent generation, so the call keeps its hands off when the
        # application froze objects itself (a pre-fork server) or another fit_blends is running.
        import gc

        with _freeze_lock:
            mine = gc.get_freeze_count() == 0 and not _freeze_users[0]
            if mine:
                gc.freeze()
            if mine or _freeze_users[0]:
                _freeze_users[0] += 1
                mine = True
        try:
            out, fit_blends.errors = _fit_blends_on(blends, device, **kw)
        finally:
            if mine:
                with _freeze_lock:
                    _freeze_users[0] -= 1
                    if not _freeze_users[0]:
                        gc.unfreeze()
        return out
    from concurrent.futures import ThreadPoolExecutor
    from .dist import shard_range

    cuts = [shard_range(len(blends), i, len(devices)) for i in range(len(devices))]
    with ThreadPoolExecutor(len(devices)) as pool:
        jobs = [pool.submit(_fit_blends_on, blends[lo:hi], int(dev), **kw)
                for (lo, hi), dev in zip(cuts, devices)]
        out = []
        for (lo, _), job in zip(cuts, jobs):
            part, errs = job.result()
            out.extend(part)
            fit_blends.errors.extend((lo + i, e) for i, e in errs)
    return out


fit_blends.errors = []


def _device_hook_covers(node):
    """True when everything ``node.update()`` can do is an ``ImageMorphology.update`` of a
    factorized component (no shift, no point source) below it -- what the device's resize test
    stands for.  Any other ``update`` in the tree (a user subclass) keeps the blend on the path
    that calls every hook on the host."""
    if isinstance(node, FactorizedComponent):
        spectrum, morphology = node.children
        return (type(node).update is FactorizedComponent.update
                and type(spectrum).update is Model.update
                and type(morphology).update is ImageMorphology.update
                and not getattr(morphology, "shifting", False))
    if isinstance(node, CombinedComponent):
        return (type(node).update is CombinedComponent.update
                and all(_device_hook_covers(c) for c in node.children))
    return False


def _next_round(local, budget):
    """Iterations until the resize hook after local iterations 10, 20, ... has to run (once
    11, 21, ... iterations of this adaprox call are done), capped by ``budget``."""
    n_hook = (11 if local == 0 else ((local - 1) // 10 + 1) * 10 + 1) - local
    return min(n_hook, budget)


def _fit_group_rebuilt(group, device, max_iter, opt, step_kw):
    """Blends that share the frame and kernel shapes, with components the resident path does
    not cover (point sources, free shifts): a device batch per round and iteration counter,
    state over the host in between."""
    while True:
        todo = [r for r in group if r.result is None and r.total < max_iter]
        if not todo:
            return
        by_local = {}
        for r in todo:
            by_local.setdefault(r.local, []).append(r)
        for local, part in by_local.items():
            comps = [_flatten(r.blend.sources) for r in part]
            n = _next_round(local, min(max_iter - r.total for r in part))
            kernel = part[0].obs[2]
            batch = BlendBatch(
                np.stack([r.obs[0] for r in part]), np.stack([r.obs[1] for r in part]),
                [r.blend._specs(c) for r, c in zip(part, comps)],
                kernel=None if kernel is None else np.stack([r.obs[2] for r in part]),
                max_iter=n, device=device)
            try:
                flat = [c for cs in comps for c in cs]
                if any(r.blend._loss_constant for r in part):
                    batch.add_loss_constant([r.blend._loss_constant for r in part])
                Blend._upload_state(batch, flat)
                batch.set_optimizer(**opt)
                if local > 0:  # the stopping rule compares with the loss before this round
                    batch.set_previous_loss(np.array([r.blend.loss[-1] for r in part]))
                batch.step(local, n, check_convergence=True, **step_kw)
                states = batch.states()
                losses = batch.loss_history()
                Blend._download_all(batch, flat)
            finally:
                batch.close()
            for r, state, loss in zip(part, states, losses):
                blend = r.blend
                blend.loss.extend(loss)
                done = len(loss)
                if state == 3:
                    r.result = ArithmeticError("parameters of the blend are not finite")
                    continue
                hook = done == n and local + done > 1 and (local + done - 1) % 10 == 0
                r.local = local + done
                restart = False
                if hook:
                    for src in blend.sources:
                        try:
                            src.update()
                        except UpdateException:
                            restart = True
                if restart:
                    r.base, r.local = len(blend.loss), 0
                elif state == 2 or r.total >= max_iter:
                    r.result = True


# fit_blends calls that share the gc.freeze() of the first of them (see there)
_freeze_lock = __import__("threading").Lock()
_freeze_users = [0]


def _device_resize_covers(blend, flat=None):
    """True when the device can also carry out every resize below ``blend``
    (``smi_batch_update_components`` with keep = 2 / 3): stock ``shrink_box``, odd square boxes,
    a constant step on float32 / float64 images.  ``flat``: the blend's components, if the
    caller has them already."""
    for c in (_flatten(blend.sources) if flat is None else flat):
        morphology = c.children[1]
        image = morphology._parameters[0]
        h, w = morphology.bbox.shape[-2:]
        if (type(morphology).shrink_box is not Morphology.shrink_box or h != w or h % 2 == 0
                or h > 1000 or image.dtype not in (np.float32, np.float64)):
            return False
        if morphology.resizing and not image.fixed and not isinstance(image.step, (int, float)):
            return False
    return True


def _standard_size(size):
    """``get_minimal_boxsize`` (initialization.py:173-177) of an array of sizes."""
    return 21 + 10 * np.ceil(np.maximum(size - 21, 0) / 10).astype(np.int64)


def _fit_group_resident(group, device, max_iter, opt, step_kw):
    """Blends that share the frame and kernel shapes, factorized image components only: ONE
    device batch for the whole fit, and every launch steps ALL blends that are still
    iterating.  The observation is uploaded once.  A blend whose boxes change starts its
    adaprox call anew (blend.py:276-302) while its batch mates go on: the device keeps the
    counter at which each blend's call began (``smi_batch_set_iteration_base``).  At a resize
    hook the device evaluates the two reductions ``ImageMorphology.update`` decides on for
    every component (``smi_batch_resize_test``); for blends made of the stock classes the
    resize itself -- centred slice, or zero-padded moments and a ``linear_ramp``-padded
    image, step halved (morphology.py:132-207) -- also happens on the device
    (``smi_batch_update_components``, keep = 2 / 3) and the Python objects learn their new
    boxes when the fit is over.  Other blends in which a box may change come to the host,
    have their sources' ``update()`` run -- the host keeps the last word -- and go back as
    new rows of the component table."""
    nb = len(group)
    comps = [_flatten(r.blend.sources) for r in group]
    specs = [r.specs for r in group]
    first = np.concatenate([[0], np.cumsum([len(c) for c in comps])]).astype(np.int64)
    kernel = group[0].obs[2]
    batch = BlendBatch(
        np.stack([r.obs[0] for r in group]), np.stack([r.obs[1] for r in group]), specs,
        kernel=None if kernel is None else np.stack([r.obs[2] for r in group]),
        max_iter=max(max_iter, 1), device=device)
    write_back, wrote = None, False
    try:
        flat = [c for cs in comps for c in cs]
        n_comp = len(flat)
        if any(r.blend._loss_constant for r in group):
            batch.add_loss_constant([r.blend._loss_constant for r in group])
        Blend._upload_state(batch, flat)
        batch.set_optimizer(**opt)
        prior = np.array([len(r.blend.loss) for r in group])  # losses of earlier fits
        base = np.zeros(nb, dtype=np.int64)
        local = np.zeros(nb, dtype=np.int64)
        count = np.zeros(nb, dtype=np.int64)  # losses recorded on the device so far
        state = np.zeros(nb, dtype=np.int32)  # 0 iterating, 2 finished, 3 failed
        frozen = np.zeros(nb, dtype=bool)  # out of iterations
        g = 0  # the batch's iteration counter: blend i is at g - (its counter base) = local[i]
        # per component: may update() act, and does the device test stand for it
        resizable = np.array([bool(c.children[1].resizing) and not c.children[1]._parameters[0].fixed
                              for c in flat])
        blend_of = np.repeat(np.arange(nb), np.diff(first))
        # blends the device resizes by itself, and what the host tracks for their components
        on_device = np.array([_device_resize_covers(r.blend, cs) for r, cs in zip(group, comps)])
        if os.environ.get("SCARLET_AMD_FIT_BLENDS") == "host-resize":  # development aid: A/B runs
            on_device[:] = False
        origin = np.array([c.children[1].bbox.origin[-2:] for c in flat], dtype=np.int64).reshape(-1, 2)
        step = np.array([float(c.children[1]._parameters[0].step)
                         if isinstance(c.children[1]._parameters[0].step, (int, float)) else np.nan
                         for c in flat])
        wide = np.array([c.children[1]._parameters[0].dtype == np.float64 for c in flat])
        moved = np.zeros(n_comp, dtype=bool)
        # (update() of a source stops at its first child that resizes, component.py:172-185)
        # (a source is one factorized component or a combined one: no second walk through the tree
        # for the common case)
        source_of = np.concatenate(
            [np.full(1 if isinstance(src, FactorizedComponent) else len(_flatten([src])), j)
             for j, src in enumerate(src for r in group for src in r.blend.sources)]).astype(np.int64) \
            if n_comp else np.zeros(0, dtype=np.int64)
        def write_back():
            """Device -> Python objects: new image Parameters (morphology.py:155-163, 180-193)
            and boxes of the components the device has resized, then all values, moments and
            the losses recorded since the fit began."""
            for k in np.flatnonzero(moved):
                morphology = flat[k].children[1]
                image = morphology._parameters[0]
                shape = tuple(batch._shapes[k])
                morphology._parameters = (
                    Parameter(np.zeros(shape, dtype=image.dtype), name=image.name, prior=image.prior,
                              constraint=image.constraint, step=float(step[k]), fixed=image.fixed),
                ) + morphology._parameters[1:]
                morphology.bbox.origin = tuple(int(o) for o in origin[k])
                morphology.bbox.shape = shape
            if moved.any():
                sources = [src for r in group for src in r.blend.sources]
                for j in np.unique(source_of[moved]):
                    _refresh_boxes(sources[j])
            Blend._download_all(batch, flat)
            history = batch.loss_history()
            n_loss = batch.progress()[1]
            for i, r in enumerate(group):
                r.blend.loss.extend(history[i][:n_loss[i]])

        pushed = None  # what the device holds as per-blend states / counter bases
        lockstep = os.environ.get("SCARLET_AMD_FIT_BLENDS") == "lockstep"
        while True:
            live = (state == 0) & ~frozen
            left = max_iter - base - local
            frozen |= live & (left <= 0)
            live &= ~frozen
            if not live.any():
                break
            # Every blend up to ITS next resize hook (after 11, 21, ... iterations of its own
            # adaprox call) or to the end of its budget: the device pauses it there
            # (smi_batch_set_pause_at) while its batch mates go on in the same launches.  Blends
            # whose calls restarted at different times no longer stop each other at every hook
            # of any of them (1024 benchmark blends: 12 rounds instead of 32).
            n_hook = np.where(local == 0, 11, ((local - 1) // 10 + 1) * 10 + 1 - local)
            quota = np.minimum(n_hook, left)
            if lockstep:  # (development aid: all blends to the nearest hook of any of them)
                quota = np.full(nb, int(quota[live].min()))
            n = int(quota[live].max())
            now_push = (np.where(frozen & (state == 0), 2, state).astype(np.int32),
                        np.where(live, g - local, 0))
            batch.set_round(
                now_push[0] if pushed is None or not np.array_equal(pushed[0], now_push[0]) else None,
                now_push[1] if pushed is None or not np.array_equal(pushed[1], now_push[1]) else None,
                np.where(live, g + quota - 1, -1))
            batch.step(g, n, check_convergence=True, **step_kw)
            g += n
            now, cnt, stopped = batch.round()
            stopped = stopped != 0
            done = cnt - count
            count = cnt.astype(np.int64)
            local[live] += done[live]
            # failed / stopped by its own rule / paused: goes on
            state[live] = np.where(now[live] == 3, 3, np.where(stopped[live], 2, 0))
            pushed = (now.astype(np.int32), now_push[1])  # (what the device holds now)
            hook = live & (now != 3) & (done == quota) & (local > 1) & ((local - 1) % 10 == 0)
            if not hook.any() or not resizable.any():
                continue
            # candidates by the device's reductions; 1e-6: the host decides what is close
            margin, pull = batch.resize_test()
            shapes = np.array(batch._shapes)
            size = shapes.max(axis=1)
            inner = size - 2 * np.where(margin == np.iinfo(np.int32).max,
                                        (shapes.min(axis=1) + 1) // 2, margin)
            standard = _standard_size(inner)
            at_hook = hook[blend_of] & resizable
            shrink = at_hook & (standard < size)
            # (the device's pull is taken with the constant step of the table: an image with a
            # step rule of its own is always shown to the host's update())
            grow = at_hook & ~shrink & ((pull > 0.1 * (1 - 1e-6)) | np.isnan(step))
            restart = np.zeros(nb, dtype=bool)
            keep = np.ones(n_comp, dtype=np.int32)
            states = []
            resized = None
            # -- blends of the stock classes: the device resizes
            dev = (shrink | grow) & on_device[blend_of]
            close = np.flatnonzero(dev & grow & (pull < 0.1 * (1 + 1e-6)))
            if close.size:  # the host's own arithmetic on the edges of these few
                for k, rec in zip(close, batch.component_states(close)):
                    edges = _edge_pull(rec["morph"], rec["m_morph"].astype(np.float64),
                                       rec["v_morph"].astype(np.float64), step[k])
                    grow[k] = dev[k] = bool(np.any(edges > 0.1))
            rows = np.flatnonzero(dev)
            if rows.size:
                rows = rows[np.unique(source_of[rows], return_index=True)[1]]  # first of its source
                new_size = np.where(shrink[rows], standard[rows], _standard_size(size[rows] + 1))
                inset = (size[rows] - new_size) // 2  # < 0: the pad of a growing box
                origin[rows] += inset[:, None]
                step[rows] /= 2
                moved[rows] = True
                keep[rows] = np.where(wide[rows], 3, 2)
                resized = dict(rows=rows, origin_y=origin[rows, 0], origin_x=origin[rows, 1],
                               size=new_size, morph_step=step[rows])
                restart[blend_of[rows]] = True
            # -- the others: only the sources with a candidate below them go over the host:
            # update() of the others is a no-op by the test above.  (Within a source the
            # reference stops at the first child that resizes: a source is visited as a whole.)
            wanted = (shrink | grow) & ~on_device[blend_of]
            visit = [int(i) for i in np.flatnonzero(hook & ~on_device)
                     if wanted[first[i]:first[i + 1]].any()]
            if visit:
                calls = []  # (blend, source, index of its first component, components)
                for i in visit:
                    k = int(first[i])
                    for src in group[i].blend.sources:
                        below = _flatten([src])
                        if wanted[k:k + len(below)].any():
                            calls.append((i, src, k, below))
                        k += len(below)
                idx = np.concatenate([np.arange(k, k + len(below)) for _, _, k, below in calls])
                for k, rec in zip(idx, batch.component_states(idx)):
                    _record_to_parameters(flat[k], rec)
                changed = set()
                images = {i: [c.children[1]._parameters[0] for c in comps[i]] for i in visit}
                for i, src, k, below in calls:
                    try:
                        src.update()
                    except UpdateException:
                        changed.add(i)
                # new rows of the component table: a component whose image Parameter was
                # replaced (sliced or padded, step halved) takes its state from the host, all
                # others keep theirs on the device
                for i in sorted(changed):
                    restart[i] = True
                    before = images[i]
                    comps[i] = _flatten(group[i].blend.sources)
                    flat[first[i]:first[i + 1]] = comps[i]
                    for j, c in enumerate(comps[i]):
                        if c.children[1]._parameters[0] is before[j]:
                            continue
                        specs[i][j] = _resized_spec(specs[i][j], c)
                        keep[first[i] + j] = 0
                        states.append(_parameters_to_record(c))
            if not restart.any():
                continue
            batch.update_components(specs, keep, states, resized=resized)
            # adaprox starts anew (a restarted blend that was about to stop goes on)
            base[restart] = prior[restart] + count[restart]
            local[restart] = 0
            state[restart & (state == 2)] = 0
        write_back()
        wrote = True
    except BaseException:
        # a launch or a host hook raised: what the device holds -- boxes it resized, parameters,
        # moments, the losses of the iterations that ran -- still reaches the Python objects, so
        # that no blend is left with a box its Parameters do not fit
        if write_back is not None and not wrote:
            try:
                write_back()
            except Exception:
                pass
        raise
    finally:
        batch.close()
    for i, r in enumerate(group):
        r.base, r.local = int(base[i]), int(local[i])
        r.result = (ArithmeticError("parameters of the blend are not finite")
                    if state[i] == 3 else True)


def _resized_spec(spec, comp):
    """The device description of a component whose box ``ImageMorphology.update`` has just
    changed: the update replaces the image Parameter by its slice or padded copy in a new box
    and halves its (constant) step (morphology.py:146-193); spectrum, constraints and everything
    else of the description stay.  Same as ``Blend._specs`` would make from scratch
    (tests/test_host_logic.py), without walking through the constraint chain again."""
    import copy

    morphology = comp.children[1]
    image = morphology._parameters[0]
    new = copy.copy(spec)
    new.morph = np.asarray(image, dtype=np.float32)
    new.origin = tuple(int(o) for o in morphology.bbox.origin[-2:])
    new.morph_step = float(image.step)
    return new


def _record_to_parameters(comp, rec):
    """A state record of the device (``BlendBatch.component_states``) into the Parameters of
    a factorized component: values in place, moments as float64 arrays."""
    sed = comp.children[0]._parameters[0]
    image = comp.children[1]._parameters[0]
    sed[...] = rec["sed"]
    sed.m, sed.v, sed.vhat = (rec[n].astype(np.float64) for n in ("m_sed", "v_sed", "vhat_sed"))
    image[...] = rec["morph"]
    image.m, image.v, image.vhat = (rec[n].astype(np.float64)
                                    for n in ("m_morph", "v_morph", "vhat_morph"))


def _parameters_to_record(comp):
    sed = comp.children[0]._parameters[0]
    image = comp.children[1]._parameters[0]
    return dict(sed=np.asarray(sed), m_sed=sed.m, v_sed=sed.v, vhat_sed=sed.vhat,
                morph=np.asarray(image), m_morph=image.m, v_morph=image.v, vhat_morph=image.vhat)


def _fit_blends_on(blends, device, max_iter=200, e_rel=1e-3, min_iter=1, _from_fit=False,
                   **alg_kwargs):
    """``fit_blends`` of ``blends`` on GPU ``device``: (results, [(index, error)]).
    ``_from_fit``: the call comes from ``Blend.fit`` itself, which keeps a blend that has to be
    fitted by its own loop (``None, None`` is returned then)."""
    if alg_kwargs.get("callback") is not None or alg_kwargs.get("scheme", "amsgrad") != "amsgrad":
        # a callback sees every blend's parameters after every iteration, another scheme of
        # proxmin.adaprox steps on the host from the device's gradients: both are Blend.fit's
        # host-stepped modes, one blend at a time -- which is what this call stands for
        # (scarlet/testing/api.py:216-224)
        out, errors = [], []
        for i, b in enumerate(blends):
            b.device = device
            try:
                out.append(b.fit(max_iter, e_rel, min_iter, **alg_kwargs))
            except ArithmeticError as e:
                errors.append((i, e))
                out.append((len(b.loss), float("nan")))
        return out, errors
    alg_kwargs.pop("scheme", None)
    alg_kwargs.pop("callback", None)
    prox_max_iter, opt = _adaprox_options(alg_kwargs)

    class _Run:
        def __init__(self, blend, obs):
            # `base + local` is the reference's `it`: 0 at the start of fit(), the length
            # of the whole loss history after a restart (blend.py:101, 198)
            self.blend, self.base, self.local, self.result = blend, 0, 0, None
            blend._scheme = ("amsgrad", 0.25)  # the batched device loop
            self.obs = obs

        @property
        def total(self):
            return self.base + self.local

    # blends with host-updated parameters (hoststep.py) step one iteration per device call
    solo, observed, described = set(), {}, {}
    for i, b in enumerate(blends):
        b._psf, b._scheme = None, ("amsgrad", 0.25)  # nothing left over from an earlier fit()
        if any(not p.fixed for obs in b.observations for p in obs.parameters):
            solo.add(i)  # free renderer parameters (psf_shift): Blend.fit's own loop
            continue
        if any(type(obs.renderer) not in (NullRenderer, ConvolutionRenderer, ResolutionRenderer)
               for obs in b.observations):
            solo.add(i)  # a user-defined renderer: Blend.fit's host-rendered mode
            continue
        observed[i] = b._observation()  # (built once: the data, weight and kernel cubes)
        if b._lowres or b._extra_layers:
            # a ResolutionRenderer observation / several observations of one channel are
            # terms of ONE blend's loss on the device (smi_batch_attach_lowres,
            # smi_batch_add_observation): such a blend is fitted by itself, like
            # [b.fit() for b in blends] would (scarlet/testing/api.py:216-224)
            solo.add(i)
    for i, b in enumerate(blends):
        if i in solo or i not in observed:
            continue
        described[i] = b._specs(_flatten(b.sources))  # (once: 15 us per component)
        if b._host:
            solo.add(i)
    if solo and _from_fit:
        return None, None
    solo_results = {}
    for i in sorted(solo):
        blends[i].device = device
        try:
            solo_results[i] = blends[i].fit(max_iter, e_rel, min_iter, prox_max_iter=prox_max_iter, **opt)
        except ArithmeticError as e:
            solo_results[i] = e
    runs = [_Run(b, observed[i]) for i, b in enumerate(blends) if i not in solo]
    for r, i in zip(runs, (i for i in range(len(blends)) if i not in solo)):
        r.specs = described[i]
    step_kw = dict(e_rel=e_rel, min_iter=min_iter, prox_max_iter=prox_max_iter)
    by_shape = {}
    for r in runs:
        data, _, kernel = r.obs
        by_shape.setdefault((data.shape, None if kernel is None else kernel.shape), []).append(r)
    for group in by_shape.values():
        plain = all(_device_hook_covers(src) for r in group for src in r.blend.sources)
        if os.environ.get("SCARLET_AMD_FIT_BLENDS") == "rebuild":  # development aid: A/B runs
            plain = False
        if plain:
            _fit_group_resident(group, device, max_iter, opt, step_kw)
        else:
            _fit_group_rebuilt(group, device, max_iter, opt, step_kw)
    out, errors = [], []
    batched = iter(runs)
    for i, blend in enumerate(blends):
        if i in solo:
            res = solo_results[i]
            if isinstance(res, Exception):
                errors.append((i, res))
                res = (len(blend.loss), float("nan"))
            out.append(res)
            continue
        r = next(batched)
        if isinstance(r.result, Exception):
            errors.append((i, r.result))
            out.append((len(blend.loss), float("nan")))
            continue
        for p in blend.parameters:
            if p.v is not None:
                p.std = STD_FROM_V
        out.append((len(blend.loss), -blend.loss[-1]))
    return out, errors
