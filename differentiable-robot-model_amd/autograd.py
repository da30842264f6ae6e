"""The differentiable nodes of the path: `torch.autograd.Function`s whose forward AND backward are launches of the HIP kernels
(csrc/, through backend.py).  What torch autograd does for the reference by recording one node per tiny tensor op of its
per-link Python loops (robot_model.py:139-195, 250-375, 402-450, 487-624 with learnable parameters robot_model.py:669-713) is
here one node per public method:

    _FkPositions      compute_forward_kinematics (positions and quaternions)          drm_fk / drm_fk_backward
    _FkJacobian       compute_endeffector_jacobian (+ the pose)                       drm_fk_jacobian / drm_fk_jacobian_backward
    _FkMse            fk_mse_loss: FK + MSE + gradients in one pass                    drm_fk_mse
    _InverseDynamics  compute_inverse_dynamics / compute_non_linear_effects           drm_rnea / drm_rnea_backward
    _MassMatrix       compute_lagrangian_inertia_matrix                               drm_crba / drm_rnea_backward per column
    _ForwardDynamics  compute_forward_dynamics (implicit differentiation)             drm_forward_dynamics / drm_rnea_backward
    _FirstOrderOnly   marks the gradients of _FkMse computed under create_graph=True (differentiating through them raises)
    _GradLaunch       a first-order gradient launch as a differentiable node (create_graph=True: second derivatives with respect to
                      q / qd / qdd / f, the output cotangents and — round 6 — the walk's table, i.e. the learnable link parameters)

plus the host-side maps between quaternion gradients and rotation-matrix adjoints (the reference's quaternion is assembled from
the entries of R by a per-sample case rule, spatial_vector_algebra.py:108-136).  robot_model.py holds the model class only.
"""
import torch

from . import backend


def _rot_from_quat(quat: torch.Tensor) -> torch.Tensor:
    """[..., 9] row-major rotation matrix of a unit quaternion (xyzw); the same for q and -q."""
    x, y, z, w = quat.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                        2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                        2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], dim=-1)


def _quat_cases(R: torch.Tensor):
    """The case of the reference's get_quaternion (spatial_vector_algebra.py:117-128) each rotation falls into — masks
    (isW, isX, isY, isZ) — and its t (the quaternion is u * 0.5 / sqrt(t))."""
    r = lambda i, j: R[..., 3 * i + j]
    tW = r(0, 0) + r(1, 1) + r(2, 2) + 1
    isW = tW > 1
    yx = r(1, 1) > r(0, 0)
    isZ = ~isW & (r(2, 2) > torch.where(yx, r(1, 1), r(0, 0)))
    isY = ~isW & ~isZ & yx
    isX = ~isW & ~isZ & ~isY
    tX = r(0, 0) - (r(1, 1) + r(2, 2)) + 1
    tY = r(1, 1) - (r(2, 2) + r(0, 0)) + 1
    tZ = r(2, 2) - (r(0, 0) + r(1, 1)) + 1
    t = torch.where(isW, tW, torch.where(isZ, tZ, torch.where(isY, tY, tX)))
    return (isW, isX, isY, isZ), t


_QUAT_CYCLES = ((0, 1, 2), (1, 2, 0), (2, 0, 1))    # (i, j, k) of the X / Y / Z cases


def _u_from_rot(R: torch.Tensor, masks) -> torch.Tensor:
    """The un-normalised quaternion u [..., 4] (xyzw) the given case assembles from the entries of R — linear in R."""
    r = lambda i, j: R[..., 3 * i + j]
    isW = masks[0]
    u = [torch.where(isW, a, torch.zeros_like(a)) for a in (r(2, 1) - r(1, 2), r(0, 2) - r(2, 0), r(1, 0) - r(0, 1),
                                                          r(0, 0) + r(1, 1) + r(2, 2) + 1)]
    for mask, (i, j, k) in zip(masks[1:], _QUAT_CYCLES):
        vals = {i: r(i, i) - (r(j, j) + r(k, k)) + 1, j: r(i, j) + r(j, i), k: r(k, i) + r(i, k), 3: r(k, j) - r(j, k)}
        for c, val in vals.items():
            u[c] = torch.where(mask, val, u[c])
    return torch.stack(u, dim=-1)


def _u_grad_to_rot(masks, gu: torch.Tensor) -> torch.Tensor:
    """dL/dR [..., 3, 3] from dL/du [..., 4]: the transpose of _u_from_rot."""
    gx, gy, gz, gw = gu.unbind(-1)
    zero = torch.zeros_like(gx)
    out = torch.zeros(gu.shape[:-1] + (9,), device=gu.device, dtype=gu.dtype)

    def scatter(mask, entries):
        for (i, j), v in entries:
            out[..., 3 * i + j] += torch.where(mask, v, zero)

    scatter(masks[0], [((2, 1), gx), ((1, 2), -gx), ((0, 2), gy), ((2, 0), -gy), ((1, 0), gz), ((0, 1), -gz),
                       ((0, 0), gw), ((1, 1), gw), ((2, 2), gw)])
    for mask, (i, j, k), (gi, gj, gk) in zip(masks[1:], _QUAT_CYCLES, ((gx, gy, gz), (gy, gz, gx), (gz, gx, gy))):
        scatter(mask, [((i, i), gi), ((j, j), -gi), ((k, k), -gi), ((i, j), gj), ((j, i), gj), ((k, i), gk), ((i, k), gk),
                       ((k, j), gw), ((j, k), -gw)])
    return out.reshape(gu.shape[:-1] + (3, 3))


def _quat_grad_to_rot(quat: torch.Tensor, grad_quat: torch.Tensor) -> torch.Tensor:
    """dL/dR [..., 3, 3] of the rotation matrix a quaternion output was taken from, given dL/dquat [..., 4] (xyzw).

    The reference's get_quaternion (spatial_vector_algebra.py:108-136) copies sums and differences of entries of R into
    the un-normalised quaternion u inside autograd and scales it by 0.5 / math.sqrt(t) — a Python float, i.e. a constant
    to autograd.  So dL/du = dL/dquat * scale and dL/dR scatters dL/du back onto the entries each case reads:
      t = tr R + 1 > 1:  u = (R21 - R12, R02 - R20, R10 - R01, t)
      else, with i the largest diagonal entry and (i, j, k) cyclic:  t = R_ii - (R_jj + R_kk) + 1,
                         u_i = t, u_j = R_ij + R_ji, u_k = R_ki + R_ik, u_w = R_kj - R_jk.
    R is rebuilt from the (unit) quaternion output; the case is re-derived from it with the reference's tests."""
    masks, t = _quat_cases(_rot_from_quat(quat))
    return _u_grad_to_rot(masks, grad_quat * (0.5 * torch.rsqrt(t)).unsqueeze(-1))


SECOND_ORDER_STEP = 4e-2    # h of the Richardson-extrapolated central differences below (radians / input units per unit direction)


class _GradLaunch(torch.autograd.Function):
    """A first-order gradient launch  g(x; c) = J(x)^T c  as a differentiable node, so that `create_graph=True` works
    through the hand-written backward kernels (gradient penalties, Hessian-vector products, Hessians row by row; the
    reference gets these from torch autograd on its tensor ops).  There is no second-order adjoint kernel: both derivatives
    of the node are DIRECTIONAL derivatives along the incoming cotangent u, taken as central differences of first-order
    launches with one Richardson step,
        d/dc:  J(x) u          = d/de fwd(x + e u)          (forward launches)
        d/dx:  d/dx <g(x;c),u> = d/de bwd(x + e u; c)       (backward launches)
    one difference quotient per input x_i (four launches for the joint angles: h = SECOND_ORDER_STEP / max|u_b| per sample with
    D = (4 D(h/2) - D(h)) / 3; two each for qd and qdd, whose step is the input's own magnitude — the outputs are at most
    quadratic in them, so the quotient is exact at any step).  ERROR MODEL: fp32 differences of first-order launches, i.e.
    an ABSOLUTE error of about 1e-6 x (scale of the first-order gradient) / h per entry — ~1e-4 of the gradient's scale,
    whatever the size of the second derivative itself: Hessian entries that are small next to the gradient are resolved only
    to that floor (tests/test_second_order.py holds the result to the reference's autograd at 2e-3 of the largest entry, also for
    |qd|, |qdd| ~ 50).  Third derivatives are not provided (this node's backward is once-differentiable).

    Round 6: one of the inputs may be the walk's constant TABLE ([cap, 32], not per-sample: `table_at` = its index among xs, -1 for
    none) — second derivatives with respect to learnable link parameters (the reference gets them from autograd on its per-link
    tensor ops, robot_model.py:669-713).  The table is differenced along the incoming cotangent of its gradient (which is non-zero
    only in the blocks a learnable parameter feeds) with ONE step for the whole table, `table_step` x the largest |entry| of the
    direction, and one Richardson step: the outputs are polynomials in the table entries of inverse dynamics / the inertia matrix /
    forward kinematics (exact for the dynamic blocks, which enter linearly), rational for forward dynamics (which passes a smaller
    step: a table whose inertias went indefinite has no accelerations).

    fwd(xs) -> tuple of outputs;  bwd(xs, cs) -> tuple of gradients, one per x;  args = xs (n_x tensors: [B, ...], the table
    [cap, 32]) then cs."""

    @staticmethod
    def forward(ctx, fwd, bwd, n_x, table_at, table_step, *args):
        xs, cs = args[:n_x], args[n_x:]
        grads = bwd(xs, cs)
        ctx.fwd, ctx.bwd, ctx.n_x, ctx.table_at, ctx.table_step = fwd, bwd, n_x, table_at, table_step
        ctx.save_for_backward(*args)
        return tuple(grads)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *us):
        args = ctx.saved_tensors
        n_x = ctx.n_x
        xs, cs = args[:n_x], args[n_x:]
        us = [u.to(torch.float32) if u is not None else torch.zeros_like(x) for u, x in zip(us, xs)]
        B = xs[0].shape[0]
        bc = lambda t, like: t.reshape((B,) + (1,) * (like.ndim - 1))
        table_at = ctx.table_at

        def directional(f, summed=None):
            """D_u f = sum_i D_{u_i} f, one difference quotient per INPUT (the directional derivative is linear in the direction)
            so that every input gets a step of its own size: x_0 (joint angles, the outputs are trigonometric in them) the
            absolute step h with one Richardson step; x_1, x_2 (qd, qdd: the outputs are polynomials of degree <= 2 in them,
            for which a central difference is EXACT at any step) a step of the input's own magnitude, max(1, max|x_i|) per
            sample — the rounding error of a difference is eps |f| / step, so a fixed small step would lose the second
            derivatives of fast, hard-accelerating states in the noise of torques that grow like qd^2.
            `summed`: the index of an output that is a SUM over the batch (the table's gradient): per-sample steps cannot be
            undone behind a sum, so that output is differenced once more with ONE step factor for the whole batch (every sample
            moves by eta u_b)."""
            total = None
            for i, (x, u) in enumerate(zip(xs, us)):
                if i == table_at:      # the constant table: one direction, one step for all of it, one Richardson step
                    top = u.abs().max()
                    if not bool(top > 0):
                        continue
                    d = u / top
                    # the step is RELATIVE to the entries the direction moves (their |d|-weighted mean size): a hand's inertias are
                    # 1e-6 kg m^2, an arm's 1e-2, a rotation's entries 1 — an absolute step fits none of them at once
                    size = float(((d.abs() * x.abs()).sum() / d.abs().sum()).clamp_min(1e-12))
                    part = None
                    for rel, weight in ((0.5 * ctx.table_step, 4.0 / 3.0), (ctx.table_step, -1.0 / 3.0)):
                        step = rel * size
                        shifted = lambda sgn: [xx + sgn * step * d if j == i else xx for j, xx in enumerate(xs)]
                        hi, lo = f(shifted(1.0)), f(shifted(-1.0))
                        term = [(a - b) * (weight / (2.0 * step)) for a, b in zip(hi, lo)]
                        part = term if part is None else [o + t for o, t in zip(part, term)]
                    part = [o * top for o in part]
                    total = part if total is None else [o + t for o, t in zip(total, part)]
                    continue
                peak = u.reshape(B, -1).abs().amax(dim=1)
                if not bool((peak > 0).any()):
                    continue
                d = u / bc(peak.clamp_min(1e-30), u)
                if i == 0:
                    plan = ((0.5 * SECOND_ORDER_STEP, 4.0 / 3.0), (SECOND_ORDER_STEP, -1.0 / 3.0))
                    size = torch.ones_like(peak)
                else:
                    plan = ((1.0, 1.0),)
                    size = x.reshape(B, -1).abs().amax(dim=1).clamp_min(1.0)
                part = None
                for step, weight in plan:
                    e = step * size
                    shifted = lambda sgn: [xx + sgn * bc(e, xx) * d if j == i else xx for j, xx in enumerate(xs)]
                    hi, lo = f(shifted(1.0)), f(shifted(-1.0))
                    term = [None if k == summed else (a - b) * bc(weight / (2.0 * e), a) for k, (a, b) in enumerate(zip(hi, lo))]
                    part = term if part is None else [None if o is None else o + t for o, t in zip(part, term)]
                part = [None if o is None else o * bc(peak, o) for o in part]
                if summed is not None:
                    top = peak.max()
                    scale = 1.0 if i == 0 else float(size.max())
                    acc = None
                    for step, weight in plan:
                        eta = step * scale / top
                        shifted = lambda sgn: [xx + sgn * eta * u if j == i else xx for j, xx in enumerate(xs)]
                        term = (f(shifted(1.0))[summed] - f(shifted(-1.0))[summed]) * (weight / (2.0 * eta))
                        acc = term if acc is None else acc + term
                    part[summed] = acc
                total = part if total is None else [o + t for o, t in zip(total, part)]
            if total is None:
                total = [torch.zeros_like(t) for t in f(list(xs))]
            return total

        need = ctx.needs_input_grad[5:]
        g_x = [None] * n_x
        g_c = [None] * len(cs)
        if any(need[:n_x]):
            got = directional(lambda X: ctx.bwd(X, cs), summed=table_at if table_at >= 0 else None)
            g_x = [g if need[i] else None for i, g in enumerate(got)]
        if any(need[n_x:]):
            got = directional(lambda X: ctx.fwd(X))
            g_c = [g if need[n_x + j] else None for j, g in enumerate(got)]
        return (None, None, None, None, None, *g_x, *g_c)


class _FkPositions(torch.autograd.Function):
    """FK of the walk's targets with a hand-written backward (csrc/drm_fk_backward.hip).

    Differentiable with respect to q and to the walk's constant table (and through its gather, to learnable ``trans`` /
    ``rot_angles`` parametrisations): the positions, and the quaternions the way the reference's are — through the
    entries of R they are assembled from, with the normalisation held constant (spatial_vector_algebra.py:108-136;
    _quat_grad_to_rot above).
    """

    @staticmethod
    def forward(ctx, q, ops_f, dw, n_targets, n_dofs, param_mask):
        pos, quat = backend.fk(dw.program, ops_f, dw.ops_i, q, n_targets, n_dofs)
        ctx.save_for_backward(q, ops_f, quat)
        ctx.dw, ctx.n_targets, ctx.n_dofs, ctx.param_mask = dw, n_targets, n_dofs, param_mask
        ctx.set_materialize_grads(False)
        return pos, quat

    @staticmethod
    def backward(ctx, grad_pos, grad_quat):
        q, ops_f, quat = ctx.saved_tensors
        want_q, want_p = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dw = ctx.dw
        if grad_pos is None:
            grad_pos = torch.zeros(quat.shape[:-1] + (3,), device=quat.device, dtype=torch.float32)
        if torch.is_grad_enabled():      # create_graph=True: the gradients are themselves a differentiable node (_GradLaunch)
            if not (want_q or want_p):
                return None, None, None, None, None, None
            T, n = ctx.n_targets, ctx.n_dofs
            const, mask = ops_f.detach(), (ctx.param_mask if want_p else 0)
            if grad_quat is None:
                grad_quat = torch.zeros_like(quat)

            # the quaternion's case and scale are constants of the reference's graph (_quat_grad_to_rot): u = A_case R
            masks, t = _quat_cases(_rot_from_quat(quat.detach()))

            def fwd(X):
                pos, qt = backend.fk(dw.program, X[1] if want_p else const, dw.ops_i, X[0], T, n)
                return pos, _u_from_rot(_rot_from_quat(qt), masks)

            def bwd(X, C):
                gq, gops = backend.fk_backward(dw.program, X[1] if want_p else const, dw.ops_i, X[0], C[0], T, n, mask, True,
                                               _u_grad_to_rot(masks, C[1]))
                return (gq.reshape(X[0].shape),) + ((gops,) if want_p else ())

            xs = [q.to(torch.float32)] + ([ops_f.to(torch.float32)] if want_p else [])
            with torch.enable_grad():
                grad_u = grad_quat.to(torch.float32) * (0.5 * torch.rsqrt(t)).unsqueeze(-1)
                got = _GradLaunch.apply(fwd, bwd, len(xs), 1 if want_p else -1, SECOND_ORDER_STEP, *xs, grad_pos.to(torch.float32), grad_u)
            return (got[0].to(q.dtype) if want_q else None), (got[1] if want_p else None), None, None, None, None
        with torch.no_grad():
            return _FkPositions._first_order(ctx, q, ops_f, quat, grad_pos, grad_quat, want_q, want_p)

    @staticmethod
    def _first_order(ctx, q, ops_f, quat, grad_pos, grad_quat, want_q, want_p):
        dw = ctx.dw
        grad_rot = _quat_grad_to_rot(quat, grad_quat.to(torch.float32)) if grad_quat is not None else None
        grad_q, grad_ops = backend.fk_backward(dw.program, ops_f, dw.ops_i, q, grad_pos, ctx.n_targets, ctx.n_dofs,
                                               ctx.param_mask if want_p else 0, want_q, grad_rot)
        if grad_q is not None:
            grad_q = grad_q.to(q.dtype).reshape(q.shape)
        return grad_q, grad_ops, None, None, None, None


class _FirstOrderOnly(torch.autograd.Function):
    """Marks a gradient that was computed under create_graph=True by a node that has no second derivatives (the fused loss node
    _FkMse): it is a correct FIRST-order gradient — trainers that always pass create_graph=True keep working — but differentiating
    THROUGH it raises instead of silently contributing a part of the answer."""

    @staticmethod
    def forward(ctx, grad):
        return grad.view_as(grad)

    @staticmethod
    def backward(ctx, _):
        raise NotImplementedError(
            "fk_mse_loss is a first-order node (forward kinematics, loss and gradients in one launch): for second derivatives "
            "compose torch.nn.functional.mse_loss(model.compute_forward_kinematics(q, link)[0], target), which is differentiable "
            "twice with respect to q and the learnable link parameters — see INTEGRATION.md, 'Second derivatives'")


class _FkMse(torch.autograd.Function):
    """loss = mean((pos(q) - target)^2) of a chain's end link with forward kinematics, loss AND gradients from one pass over q
    (backend.fk_mse, csrc/drm_fk_backward.hip MSE form): the forward call already holds d loss / d q and d loss / d ops_f, the
    backward scales them by the incoming gradient.  First order only."""

    @staticmethod
    def forward(ctx, q, target, ops_f, dw, n_dofs, param_mask):
        want_q, want_p = q.requires_grad, ops_f.requires_grad
        loss, grad_q, grad_ops = backend.fk_mse(dw.program, ops_f, dw.ops_i, q, target, n_dofs, param_mask if want_p else 0, want_q)
        ctx.save_for_backward(*[g for g in (grad_q, grad_ops) if g is not None])
        ctx.have = (grad_q is not None, grad_ops is not None)
        ctx.q_shape, ctx.q_dtype = q.shape, q.dtype
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        saved = list(ctx.saved_tensors)
        grad_q = saved.pop(0) if ctx.have[0] else None
        grad_ops = saved.pop(0) if ctx.have[1] else None
        second = torch.is_grad_enabled()          # create_graph=True: first-order gradients that say so when differentiated again
        with torch.no_grad():
            if grad_q is not None:
                grad_q = (grad_q * grad_loss).to(ctx.q_dtype).reshape(ctx.q_shape)
            if grad_ops is not None:
                grad_ops = grad_ops * grad_loss
        if second:
            with torch.enable_grad():
                grad_q = _FirstOrderOnly.apply(grad_q.requires_grad_(True)) if grad_q is not None else None
                grad_ops = _FirstOrderOnly.apply(grad_ops.requires_grad_(True)) if grad_ops is not None else None
        return grad_q, None, grad_ops, None, None, None


class _FkMseLinks(torch.autograd.Function):
    """_FkMse for a model WITH learnable links, from the links' parameter tensors to their gradients (backend.fk_mse_links,
    drm_fk_mse_links): the walk table is built inside the first launch and the gradient is taken back through that map inside the
    second, so a training step's forward + loss + backward is two launches and this ONE autograd node — instead of WalkTable (a cat and
    a launch), _FkMse (two launches) and WalkTable's backward (a launch).  ``pieces``: six tensors per learnable link, as for
    backend.WalkTable.  First order only."""

    @staticmethod
    def forward(ctx, q, target, base, sel, gsign, dw, n_dofs, param_mask, *pieces):
        want_q = q.requires_grad
        loss, grad_q, grad_params = backend.fk_mse_links(dw.program, base, dw.ops_i, sel, gsign, pieces, q, target, n_dofs, param_mask, want_q)
        ctx.save_for_backward(grad_params, *([grad_q] if want_q else []))
        ctx.want_q, ctx.q_shape, ctx.q_dtype = want_q, q.shape, q.dtype
        ctx.shapes = [tuple(p.shape) for p in pieces]
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        saved = ctx.saved_tensors
        second = torch.is_grad_enabled()
        with torch.no_grad():
            flat = (saved[0] * grad_loss).reshape(-1)      # ONE kernel for every parameter's gradient; the pieces are views of it
            grad_q = (saved[1] * grad_loss).to(ctx.q_dtype).reshape(ctx.q_shape) if ctx.want_q else None
        if second:      # create_graph=True: first-order gradients that say so when differentiated again (as _FkMse's)
            with torch.enable_grad():
                flat = _FirstOrderOnly.apply(flat.requires_grad_(True))
                grad_q = _FirstOrderOnly.apply(grad_q.requires_grad_(True)) if grad_q is not None else None
        out, off = [], 0
        for i, shape in enumerate(ctx.shapes):
            n = 1
            for d in shape:
                n *= d
            out.append(flat[off:off + n].reshape(shape) if ctx.needs_input_grad[8 + i] else None)
            off += n
        return (grad_q, None, None, None, None, None, None, None) + tuple(out)


class _FkJacobian(torch.autograd.Function):
    """Fused FK + geometric Jacobian with a hand-written backward (csrc/drm_fk_backward.hip, JAC form): position and
    both Jacobians are differentiable with respect to q and to the walk's constant table (learnable ``trans`` /
    ``rot_angles``), as torch autograd makes them in the reference (robot_model.py:626-667); so is the quaternion, through
    the entries of R it is assembled from (_quat_grad_to_rot)."""

    @staticmethod
    def forward(ctx, q, ops_f, dw, n_dofs, param_mask):
        pos, quat, lin, ang = backend.fk_jacobian(dw.program, ops_f, dw.ops_i, q, n_dofs)
        ctx.save_for_backward(q, ops_f, quat)
        ctx.dw, ctx.n_dofs, ctx.param_mask = dw, n_dofs, param_mask
        ctx.set_materialize_grads(False)
        return pos, quat, lin, ang

    @staticmethod
    def backward(ctx, grad_pos, grad_quat, grad_lin, grad_ang):
        q, ops_f, quat = ctx.saved_tensors
        want_q, want_p = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dw = ctx.dw
        zeros = lambda: torch.zeros(q.shape[0], 3, ctx.n_dofs, device=q.device, dtype=torch.float32)
        grad_lin = grad_lin if grad_lin is not None else zeros()
        grad_ang = grad_ang if grad_ang is not None else zeros()
        if torch.is_grad_enabled():      # create_graph=True: the gradients are themselves a differentiable node (_GradLaunch)
            if not (want_q or want_p):
                return None, None, None, None, None
            n = ctx.n_dofs
            const, mask = ops_f.detach(), (ctx.param_mask if want_p else 0)
            f32 = lambda t, like: (t if t is not None else torch.zeros_like(like)).to(torch.float32)
            pos_like = torch.zeros(quat.shape[:-1] + (3,), device=quat.device, dtype=torch.float32)

            masks, t = _quat_cases(_rot_from_quat(quat.detach()))     # (constants of the reference's graph, as in _FkPositions)

            def fwd(X):
                pos, qt, lin, ang = backend.fk_jacobian(dw.program, X[1] if want_p else const, dw.ops_i, X[0], n)
                return pos, _u_from_rot(_rot_from_quat(qt), masks), lin, ang

            def bwd(X, C):
                gq, gops = backend.fk_jacobian_backward(dw.program, X[1] if want_p else const, dw.ops_i, X[0], C[0], C[2], C[3], n, mask,
                                                        True, _u_grad_to_rot(masks, C[1]))
                return (gq.reshape(X[0].shape),) + ((gops,) if want_p else ())

            xs = [q.to(torch.float32)] + ([ops_f.to(torch.float32)] if want_p else [])
            with torch.enable_grad():
                grad_u = f32(grad_quat, quat) * (0.5 * torch.rsqrt(t)).unsqueeze(-1)
                got = _GradLaunch.apply(fwd, bwd, len(xs), 1 if want_p else -1, SECOND_ORDER_STEP, *xs, f32(grad_pos, pos_like), grad_u,
                                        grad_lin.to(torch.float32), grad_ang.to(torch.float32))
            return (got[0].to(q.dtype) if want_q else None), (got[1] if want_p else None), None, None, None
        with torch.no_grad():
            return _FkJacobian._first_order(ctx, q, ops_f, quat, grad_pos, grad_quat, grad_lin, grad_ang, want_q, want_p)

    @staticmethod
    def _first_order(ctx, q, ops_f, quat, grad_pos, grad_quat, grad_lin, grad_ang, want_q, want_p):
        dw = ctx.dw
        grad_rot = _quat_grad_to_rot(quat, grad_quat.to(torch.float32)) if grad_quat is not None else None
        grad_q, grad_ops = backend.fk_jacobian_backward(dw.program, ops_f, dw.ops_i, q, grad_pos, grad_lin, grad_ang,
                                                        ctx.n_dofs, ctx.param_mask if want_p else 0, want_q, grad_rot)
        if grad_q is not None:
            grad_q = grad_q.to(q.dtype).reshape(q.shape)
        return grad_q, grad_ops, None, None, None


class _InverseDynamics(torch.autograd.Function):
    """RNEA with a hand-written backward (csrc/drm_rnea_backward.hip): torques are differentiable with respect
    to q, qd, qdd and to the walk's constant table (and through its gather, to every learnable link parameter)."""

    @staticmethod
    def forward(ctx, q, qd, qdd, ops_f, dw, gravity, damping, n_dofs, param_mask):
        tau = backend.rnea(dw.program, ops_f, dw.ops_i, q, qd, qdd, gravity, damping, n_dofs)
        ctx.save_for_backward(q, qd, qdd if qdd is not None else q.new_empty(0), ops_f)
        ctx.has_qdd = qdd is not None
        ctx.dw, ctx.flags, ctx.n_dofs, ctx.param_mask = dw, (gravity, damping), n_dofs, param_mask
        return tau

    @staticmethod
    def backward(ctx, grad_tau):
        q, qd, qdd, ops_f = ctx.saved_tensors
        qdd = qdd if ctx.has_qdd else None
        want_in = any(ctx.needs_input_grad[:3])
        dw = ctx.dw
        if torch.is_grad_enabled():      # create_graph=True: the gradients are themselves a differentiable node (_GradLaunch)
            want_p = bool(ctx.needs_input_grad[3]) and ctx.param_mask != 0
            if not (want_in or want_p):
                return (None,) * 9
            const, n, (gravity, damping) = ops_f.detach(), ctx.n_dofs, ctx.flags
            has_qdd, mask = ctx.has_qdd, (ctx.param_mask if want_p else 0)
            n_in = 3 if has_qdd else 2

            def fwd(X):
                return (backend.rnea(dw.program, X[n_in] if want_p else const, dw.ops_i, X[0], X[1], X[2] if has_qdd else None, gravity,
                                     damping, n),)

            def bwd(X, C):
                gin, gops = backend.rnea_backward(dw.program, X[n_in] if want_p else const, dw.ops_i, X[0], X[1],
                                                  X[2] if has_qdd else None, C[0], gravity, damping, n, mask, True)
                return tuple(g.reshape(X[0].shape) for g in gin[:n_in]) + ((gops,) if want_p else ())

            xs = [q.to(torch.float32), qd.to(torch.float32)] + ([qdd.to(torch.float32)] if has_qdd else []) + \
                ([ops_f.to(torch.float32)] if want_p else [])
            with torch.enable_grad():
                got = _GradLaunch.apply(fwd, bwd, len(xs), n_in if want_p else -1, SECOND_ORDER_STEP, *xs, grad_tau.to(torch.float32))
            gq = got[0].to(q.dtype) if ctx.needs_input_grad[0] else None
            gqd = got[1].to(qd.dtype) if ctx.needs_input_grad[1] else None
            gqdd = got[2].to(qdd.dtype) if (has_qdd and ctx.needs_input_grad[2]) else None
            return gq, gqd, gqdd, (got[n_in] if want_p else None), None, None, None, None, None
        with torch.no_grad():
            return _InverseDynamics._first_order(ctx, q, qd, qdd, ops_f, grad_tau, want_in)

    @staticmethod
    def _first_order(ctx, q, qd, qdd, ops_f, grad_tau, want_in):
        dw = ctx.dw
        gin, grad_ops = backend.rnea_backward(dw.program, ops_f, dw.ops_i, q, qd, qdd, grad_tau, ctx.flags[0],
                                              ctx.flags[1], ctx.n_dofs, ctx.param_mask if ctx.needs_input_grad[3] else 0,
                                              want_in)
        gq = gqd = gqdd = None
        if gin is not None:
            gq = gin[0].to(q.dtype).reshape(q.shape) if ctx.needs_input_grad[0] else None
            gqd = gin[1].to(qd.dtype).reshape(qd.shape) if ctx.needs_input_grad[1] else None
            gqdd = gin[2].to(qdd.dtype).reshape(qdd.shape) if (ctx.has_qdd and ctx.needs_input_grad[2]) else None
        return gq, gqd, gqdd, grad_ops, None, None, None, None, None


class _MassMatrix(torch.autograd.Function):
    """Joint-space inertia matrix with a backward built on the reference's own definition of H
    (robot_model.py:402-450): column j is the inverse dynamics of a unit acceleration of joint j at rest without
    gravity, H[:, :, j] = ID(q, 0, e_j), so for a loss gradient G on H the gradients with respect to q and to the
    learnable link parameters are those of the RNEA with qdd = e_j and grad_tau = G[:, :, j], summed over j — one launch of
    the RNEA backward kernel over the batch stacked n times.  Under create_graph=True that gradient is a differentiable node
    (_GradLaunch: second derivatives with respect to q, the learnable parameters and the incoming cotangent)."""

    @staticmethod
    def forward(ctx, q, ops_f, dw, n_dofs, param_mask):
        H = backend.crba(dw.program, ops_f, dw.ops_i, q, n_dofs)
        ctx.save_for_backward(q, ops_f)
        ctx.dw, ctx.n_dofs, ctx.param_mask = dw, n_dofs, param_mask
        return H

    @staticmethod
    def _first_order(dw, n, qf, table, G, want_q, mask):
        """(grad_q [B, n] or None, grad_ops_f [cap, 32] or None) for the loss gradient G [B, n, n] on H(q; table)."""
        B = qf.shape[0]
        # all columns in ONE launch of the RNEA backward kernel: the batch is stacked n times (rows j B .. (j + 1) B - 1 carry
        # qdd = e_j and grad_tau = G[:, :, j]); the kernel's fixed-order reduction sums the parameter gradients over all of
        # them.  Very large batches go in groups of columns (at most 2^22 stacked rows per launch).
        group = max(1, min(n, (1 << 22) // max(B, 1)))
        gq = grad_ops = None
        eye = torch.eye(n, device=qf.device, dtype=torch.float32)
        for j0 in range(0, n, group):
            cols = min(group, n - j0)
            qs = qf.repeat(cols, 1)
            unit = eye[j0:j0 + cols].repeat_interleave(B, dim=0)
            gt = G[:, :, j0:j0 + cols].permute(2, 0, 1).reshape(cols * B, n).contiguous()
            gin, gops = backend.rnea_backward(dw.program, table, dw.ops_i, qs, torch.zeros_like(qs), unit, gt, False, False, n,
                                              mask, want_q)
            if gin is not None:
                part = gin[0].reshape(cols, B, n).sum(dim=0)
                gq = part if gq is None else gq + part
            if gops is not None:
                grad_ops = gops if grad_ops is None else grad_ops + gops
        return gq, grad_ops

    @staticmethod
    def backward(ctx, grad_H):
        q, ops_f = ctx.saved_tensors
        dw, n = ctx.dw, ctx.n_dofs
        want_q = ctx.needs_input_grad[0]
        want_p = bool(ctx.needs_input_grad[1]) and ctx.param_mask != 0
        if not (want_q or want_p):
            return None, None, None, None, None
        if torch.is_grad_enabled():      # create_graph=True
            const, mask = ops_f.detach(), (ctx.param_mask if want_p else 0)

            def fwd(X):
                # H by the reference's own definition, column j = ID(q, 0, e_j) without gravity (robot_model.py:402-450) — the function
                # the backward above differentiates.  (The composite-rigid-body kernel reads an inertia matrix as the symmetric matrix
                # it physically is; a table moved along an unsymmetric direction — an UnconstrainedTensor(3, 3) parametrisation —
                # is still the reference's function only through the RNEA.)
                x, table = X[0], (X[1] if want_p else const)
                Bx = x.shape[0]
                unit = torch.eye(n, device=x.device, dtype=torch.float32).repeat_interleave(Bx, dim=0)
                xs_ = x.repeat(n, 1)
                tau = backend.rnea(dw.program, table, dw.ops_i, xs_, torch.zeros_like(xs_), unit, False, False, n)
                return (tau.reshape(n, Bx, n).permute(1, 2, 0).contiguous(),)

            def bwd(X, C):
                gq, gops = _MassMatrix._first_order(dw, n, X[0], X[1] if want_p else const, C[0], True, mask)
                return (gq.reshape(X[0].shape),) + ((gops,) if want_p else ())

            xs = [q.to(torch.float32)] + ([ops_f.to(torch.float32)] if want_p else [])
            with torch.enable_grad():
                got = _GradLaunch.apply(fwd, bwd, len(xs), 1 if want_p else -1, SECOND_ORDER_STEP, *xs, grad_H.to(torch.float32))
            return (got[0].to(q.dtype) if want_q else None), (got[1] if want_p else None), None, None, None
        with torch.no_grad():
            gq, grad_ops = _MassMatrix._first_order(dw, n, q.to(torch.float32), ops_f, grad_H.to(torch.float32), want_q,
                                                    ctx.param_mask if want_p else 0)
        return (gq.to(q.dtype).reshape(q.shape) if gq is not None else None), grad_ops, None, None, None


FD_TABLE_STEP = 5e-2       # (relative, as SECOND_ORDER_STEP) forward dynamics is RATIONAL in the table (H^-1): the inertias stay positive definite


class _ForwardDynamics(torch.autograd.Function):
    """Forward dynamics with an implicit-function backward: qdd solves ID(q, qd, qdd; theta) = f, so for a loss
    gradient g on qdd,  lambda = H(q)^-1 g  (one more solve, the same kernel with zero bias) and

        dL/df = lambda,    dL/d(q, qd, theta) = -lambda^T dID/d(q, qd, theta) at (q, qd, qdd),

    which is exactly the RNEA backward kernel fed with grad_tau = lambda (what torch autograd gets by differentiating
    through the reference's articulated-body recursion, robot_model.py:487-624; examples/learn_forward_dynamics_iiwa.py).
    Under create_graph=True that gradient is a differentiable node (_GradLaunch)."""

    @staticmethod
    def forward(ctx, q, qd, f, ops_f, dw, gravity, damping, n_dofs, param_mask):
        qdd = backend.forward_dynamics(dw.program, ops_f, dw.ops_i, q, qd, f, gravity, damping, n_dofs)
        ctx.save_for_backward(q, qd, f, qdd, ops_f)
        ctx.dw, ctx.flags, ctx.n_dofs, ctx.param_mask = dw, (gravity, damping), n_dofs, param_mask
        return qdd

    @staticmethod
    def _first_order(dw, n, flags, q, qd, qdd, table, grad_qdd, want_in, mask):
        """(grad_q, grad_qd, grad_f, grad_ops_f) of a loss gradient on qdd = FD(q, qd, f; table) at the given solution qdd."""
        lam = backend.forward_dynamics(dw.program, table, dw.ops_i, q, torch.zeros_like(qd), grad_qdd.contiguous(), False, False, n)
        gq = gqd = grad_ops = None
        if want_in or mask:
            gin, gops = backend.rnea_backward(dw.program, table, dw.ops_i, q, qd, qdd, lam, flags[0], flags[1], n, mask, want_in)
            if gin is not None:
                gq, gqd = -gin[0].reshape(q.shape), -gin[1].reshape(qd.shape)
            grad_ops = -gops if gops is not None else None
        return gq, gqd, lam.reshape(grad_qdd.shape), grad_ops

    @staticmethod
    def backward(ctx, grad_qdd):
        q, qd, f, qdd, ops_f = ctx.saved_tensors
        dw, n = ctx.dw, ctx.n_dofs
        want_p = bool(ctx.needs_input_grad[3]) and ctx.param_mask != 0
        if torch.is_grad_enabled():      # create_graph=True
            const, mask, flags = ops_f.detach(), (ctx.param_mask if want_p else 0), ctx.flags

            def table_of(X):
                return X[3] if want_p else const

            def fwd(X):
                return (backend.forward_dynamics(dw.program, table_of(X), dw.ops_i, X[0], X[1], X[2], flags[0], flags[1], n),)

            def bwd(X, C):
                acc = backend.forward_dynamics(dw.program, table_of(X), dw.ops_i, X[0], X[1], X[2], flags[0], flags[1], n)
                gq, gqd, gf, gops = _ForwardDynamics._first_order(dw, n, flags, X[0], X[1], acc, table_of(X), C[0], True, mask)
                return (gq, gqd, gf) + ((gops,) if want_p else ())

            xs = [t.to(torch.float32) for t in (q, qd, f)] + ([ops_f.to(torch.float32)] if want_p else [])
            with torch.enable_grad():
                got = _GradLaunch.apply(fwd, bwd, len(xs), 3 if want_p else -1, FD_TABLE_STEP, *xs, grad_qdd.to(torch.float32))
            pick = lambda i, like: got[i].to(like.dtype) if ctx.needs_input_grad[i] else None
            return pick(0, q), pick(1, qd), pick(2, f), (got[3] if want_p else None), None, None, None, None, None, None
        with torch.no_grad():
            gq, gqd, gf, grad_ops = _ForwardDynamics._first_order(dw, n, ctx.flags, q, qd, qdd, ops_f, grad_qdd.to(torch.float32),
                                                                  any(ctx.needs_input_grad[:2]), ctx.param_mask if want_p else 0)
        return (gq if ctx.needs_input_grad[0] else None), (gqd if ctx.needs_input_grad[1] else None), \
            (gf if ctx.needs_input_grad[2] else None), grad_ops, None, None, None, None, None, None
