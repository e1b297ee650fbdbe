"""Model-step fallback for boxes without PyG / torch_sparse: the reference's flagship
``SAGE`` (driver/models.py:19-56: ``num_layers`` x SAGEConv(bias=False, mean aggregation), ReLU +
dropout 0.5 between layers, log_softmax) on the MFG's CSR, with the message passing on HIP kernels
(csrc/aggregate.hip) and the linear layers on the library GEMMs torch dispatches to.

The constructor, ``reset_parameters`` and ``forward(x, adjs)`` follow the reference; ``adjs`` is
what the data path delivers: ``[(adj_t, e_id, (S, T)), ...]``, outermost hop first."""
import ctypes as C

import torch
import torch.nn.functional as F

from . import _native as nat
from .fast_sampler import RowRefs, TableRows


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else None


def _stream():
    # the raw handle of the current stream of the current device (torch.cuda.current_stream() builds a Stream object: 5 us)
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


class _MeanAggregate(torch.autograd.Function):
    """out[t] = mean_{e in row t} x[col[e]]  (empty rows give 0, as PyG's mean aggregation).

    With ``concat_target`` the result is the fused operand ``[mean | x_target]`` of shape [T, 2F] (fp32)
    where x_target = x[:T] (the MFG contract, driver/models.py:44-45): both halves are written by the
    one kernel, so lin_l and lin_r become ONE GEMM, and the backward writes grad_x completely (target
    half's gradient, zeros, scattered mean gradient) instead of autograd zero-padding the slice's
    gradient and adding two full-size tensors."""

    @staticmethod
    def forward(ctx, x, rowptr, col, num_targets, concat_target):
        L = nat.load()
        nat.require_device()
        assert x.is_cuda and x.dim() == 2 and x.stride(1) == 1 and x.dtype in (torch.float16, torch.float32)
        Fdim = x.size(1)
        width = 2 * Fdim if concat_target else Fdim
        out = torch.empty((num_targets, width), dtype=torch.float32, device=x.device)
        fn = L.spp_sage_operand_forward if concat_target else L.spp_csr_mean_forward
        nat.check(fn(_p(rowptr), _p(col), num_targets, _p(x), int(x.dtype == torch.float16),
                     x.stride(0) if x.size(0) > 1 else Fdim, Fdim, _p(out), width, _stream()))
        ctx.save_for_backward(rowptr, col)
        ctx.shape = (x.size(0), Fdim, num_targets, width)
        ctx.in_dtype = x.dtype
        ctx.concat = bool(concat_target)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        rowptr, col = ctx.saved_tensors
        S, Fdim, T, width = ctx.shape
        grad_x = None
        if ctx.needs_input_grad[0]:
            g = grad_out if (grad_out.stride(1) == 1 and grad_out.dtype == torch.float32) else \
                grad_out.contiguous().to(torch.float32)
            L = nat.load()
            go_stride = g.stride(0) if T > 1 else width
            if ctx.concat and Fdim % 4 == 0 and go_stride % 4 == 0 and g.data_ptr() % 16 == 0:
                grad_x = torch.empty((S, Fdim), dtype=torch.float32, device=g.device)
                E = col.numel()
                if E * Fdim >= (1 << 22):
                    # gather over the transposed hop (built here: count, scan, fill) instead of E x F fp32 atomics
                    nbytes = int(L.spp_sage_operand_backward_workspace_bytes(T, S, E))
                    ws = torch.empty(nbytes, dtype=torch.uint8, device=g.device)
                    nat.check(L.spp_sage_operand_backward_gather(_p(rowptr), _p(col), T, S, E, _p(g), go_stride, Fdim,
                                                                 _p(grad_x), _p(ws), nbytes, _stream()))
                else:
                    nat.check(L.spp_sage_operand_backward(_p(rowptr), _p(col), T, S, _p(g), go_stride, Fdim,
                                                          _p(grad_x), _stream()))
            else:
                grad_x = torch.zeros((S, Fdim), dtype=torch.float32, device=g.device)
                nat.check(L.spp_csr_mean_backward(_p(rowptr), _p(col), T, _p(g), go_stride, Fdim, _p(grad_x), _stream()))
                if ctx.concat:
                    grad_x[:T] += g[:, Fdim:]
            grad_x = grad_x.to(ctx.in_dtype)
        return grad_x, None, None, None, None


class _ReluDropout(torch.autograd.Function):
    """F.dropout(F.relu(x), p, training) (driver/models.py:47-48) in one pass over x, with a backward
    that needs only the output (csrc/aggregate.hip k_relu_dropout_*)."""

    @staticmethod
    def forward(ctx, x, p, training):
        L = nat.load()
        xc = x.contiguous()
        y = torch.empty_like(xc)
        # the seed comes from torch's CPU generator, so torch.manual_seed makes a run repeatable
        seed = int(torch.empty((), dtype=torch.int64).random_().item()) & 0xFFFFFFFFFFFFFFFF if training else 0
        nat.check(L.spp_relu_dropout_forward(_p(xc), xc.numel(), float(p), int(bool(training)), seed, _p(y), _stream()))
        ctx.save_for_backward(y)
        ctx.scale = 1.0 / (1.0 - float(p)) if training else 1.0
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        gc = g.contiguous()
        gx = torch.empty_like(gc)
        nat.check(nat.load().spp_relu_dropout_backward(_p(gc), _p(y), gc.numel(), ctx.scale, _p(gx), _stream()))
        return gx, None, None


def relu_dropout(x, p=0.5, training=True):
    """relu followed by dropout; fp32 CUDA tensors take the fused HIP kernel"""
    if x.is_cuda and x.dtype == torch.float32 and x.numel() > 0 and x.data_ptr() % 16 == 0:
        return _ReluDropout.apply(x, p, training)
    return F.dropout(F.relu(x), p=p, training=training)


def mean_aggregate(x, rowptr, col, num_targets):
    return _MeanAggregate.apply(x, rowptr, col, num_targets, False)


class _TallLinear(torch.autograd.Function):
    """y = a @ w.T for a very tall ``a`` ([T, K], T ~ 1e5) and a small ``w`` ([N, K]).

    The weight gradient ``g.T @ a`` is a [N, K] output reduced over T: as one GEMM it has a handful
    of output tiles for 256 CUs (measured 505 us at T=165k, K=200, N=256 -- 27 % of the step).  It is
    computed as a batched GEMM over row slabs (split-K) and summed instead."""

    @staticmethod
    def forward(ctx, a, w):
        ctx.save_for_backward(a, w)
        return a @ w.t()

    @staticmethod
    def backward(ctx, g):
        a, w = ctx.saved_tensors
        grad_a = g @ w if ctx.needs_input_grad[0] else None
        grad_w = None
        if ctx.needs_input_grad[1]:
            T = a.size(0)
            slabs = min(64, T // 4096)
            if slabs >= 2 and a.is_contiguous() and g.is_contiguous():
                c = T // slabs
                head = torch.bmm(g[:slabs * c].view(slabs, c, -1).transpose(1, 2), a[:slabs * c].view(slabs, c, -1))
                grad_w = head.sum(0)
                if slabs * c < T:
                    grad_w = grad_w + g[slabs * c:].t() @ a[slabs * c:]
            else:
                grad_w = g.t() @ a
        return grad_a, grad_w


def init_weights(m):                                     # driver/models.py:12-16
    if isinstance(m, torch.nn.Linear):
        torch.nn.init.xavier_uniform_(m.weight, gain=torch.nn.init.calculate_gain("relu"))


class SAGEConv(torch.nn.Module):
    """torch_geometric.nn.SAGEConv(in, out, aggr='mean', root_weight=True, bias=...) on a bipartite
    ((x, x_target), adj_t):  lin_l(mean_j x_j) + lin_r(x_target); only lin_l carries the bias."""

    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__()
        self.lin_l = torch.nn.Linear(in_channels, out_channels, bias=bias)
        self.lin_r = torch.nn.Linear(in_channels, out_channels, bias=False)

    def reset_parameters(self):
        self.lin_l.reset_parameters()
        self.lin_r.reset_parameters()

    def forward(self, x_pair, adj_t):
        x, x_target = x_pair
        rowptr, col, _ = adj_t.csr()
        # [mean_j x_j | x_target] @ [W_l | W_r]^T: one GEMM instead of two plus an add
        T = x_target.size(0)
        if x_target.data_ptr() == x.data_ptr() and x_target.size(1) == x.size(1) and x_target.stride() == x.stride():
            fused = _MeanAggregate.apply(x, rowptr, col, T, True)           # x_target = x[:T] (the MFG contract)
        else:                                                               # a foreign target matrix
            fused = torch.cat([_MeanAggregate.apply(x, rowptr, col, T, False), x_target.to(torch.float32)], dim=1)
        out = _TallLinear.apply(fused, torch.cat([self.lin_l.weight, self.lin_r.weight], dim=1))
        return out if self.lin_l.bias is None else out + self.lin_l.bias


class SAGE(torch.nn.Module):
    def __init__(self, in_channels, hidden_channels, out_channels, num_layers):
        super().__init__()
        self.num_layers = num_layers
        self.hidden_channels = hidden_channels
        self.convs = torch.nn.ModuleList()
        self.convs.append(SAGEConv(in_channels, hidden_channels, bias=False))
        for _ in range(num_layers - 2):
            self.convs.append(SAGEConv(hidden_channels, hidden_channels, bias=False))
        self.convs.append(SAGEConv(hidden_channels, out_channels, bias=False))
        self.reset_parameters()

    def reset_parameters(self):
        for conv in self.convs:
            conv.reset_parameters()
            conv.apply(init_weights)

    def forward(self, x, adjs):
        # the reference converts the whole feature matrix to fp32 first (models.py:43); the first
        # aggregation reads the fp16 rows directly instead (exact) and only the targets are converted
        if x.is_cuda and _SageStack.usable(self, x, adjs):
            hops = []
            for adj_t, _e_id, size in adjs:
                rowptr, col, _ = adj_t.csr()
                hops.append((rowptr, col, int(size[1])))
            weights = [w for conv in self.convs for w in (conv.lin_l.weight, conv.lin_r.weight)]
            if isinstance(x, TableRows):                 # fused first layer: the batch's rows are read from the table
                x = (x.table, x.n_id)
            elif isinstance(x, RowRefs):                 # ... or wherever the partitioned path found them
                x = _Refs(x)
            return _SageStack.apply(x, hops, self.training, 0.5, *weights)
        if isinstance(x, (TableRows, RowRefs)):
            x = x.materialize()
        for i, (adj_t, _e_id, size) in enumerate(adjs):
            x_target = x[:size[1]]
            x = self.convs[i]((x, x_target), adj_t)
            if i != self.num_layers - 1:
                x = relu_dropout(x, 0.5, self.training)       # F.relu + F.dropout(p=0.5) (models.py:47-48)
        return torch.log_softmax(x, dim=-1)


def _wgrad(g, a):
    """g.T @ a for very tall g [T, N], a [T, K]: batched over row slabs and summed (see _TallLinear).  64 slabs of
    >= 2048 rows in the a.T @ g order are what the library runs fastest at T = 164 k, N = K = 256 (162 us against 174-181
    for 32 slabs and 483 for the single product; tools/gemm_variants.py)"""
    T = a.size(0)
    slabs = min(64, T // 2048)
    if slabs < 2:
        return g.t() @ a
    c = T // slabs
    acc = torch.bmm(a[:slabs * c].view(slabs, c, -1).transpose(1, 2), g[:slabs * c].view(slabs, c, -1)).sum(0)   # [K, N]
    if slabs * c < T:      # the < `slabs` rows left over: accumulated in place (one launch; it was a product + an add)
        acc.addmm_(a[slabs * c:].t(), g[slabs * c:])
    return acc.t()


def _split_wgrad(gW):
    """[N, 2K] weight gradient of the [W_l | W_r] operand -> (grad W_l, grad W_r), each [N, K] contiguous, with ONE copy
    launch for both: gW is the transposed view of a contiguous [2K, N] sum (see _wgrad), so its two halves are the two
    [K, N] blocks of that buffer, transposed together."""
    N, K2 = gW.shape
    base = gW.t()
    if not base.is_contiguous():
        return gW[:, :K2 // 2].contiguous(), gW[:, K2 // 2:].contiguous()
    both = base.view(2, K2 // 2, N).transpose(1, 2).contiguous()      # [2, N, K]
    return both[0], both[1]


def _tall_linear(a, w):
    """a @ w.T for a very tall a [T, K].  Round 3 issued it in four row chunks (206 against 214-236 us for the single
    product at T = 164 k, N = K = 256 then); with this stack's library F.linear(a, w) takes the transposed weight as it is and
    runs as fast (201-204 against 207 us, tools/corun_gemm.py), and one launch instead of an empty + a transposed copy +
    four products is 0.1 ms less host time per step: resident step 1.010-1.015 -> 0.995-0.997 ms, with the data path
    1.140-1.156 -> 1.111-1.121 (tools/overlap_ab.py, SPP_SAGE_ONE_PRODUCT A/B of round 4)."""
    return torch.nn.functional.linear(a, w)


class _Refs:
    """RowRefs on their way through _SageStack.apply (a non-tensor argument)"""
    __slots__ = ("r",)

    def __init__(self, r):
        self.r = r


class _SageStack(torch.autograd.Function):
    """The whole SAGE forward (all layers: fused operand, one GEMM, ReLU + dropout; log_softmax) as ONE
    autograd node with a hand-written backward.  The kernels are the ones the layer-wise path uses; what
    goes away is the host side: ~25 autograd nodes, nine Python-level backward calls and the slice /
    accumulate bookkeeping of x[:T] cost more wall time than the GPU work they enqueue (1.06 ms of
    kernels in a 1.34 ms step at papers scale)."""

    @staticmethod
    def usable(model, x, adjs):
        if isinstance(x, TableRows):                     # (resident table, n_id): the first layer reads the table itself
            x = x.table
        if isinstance(x, RowRefs):                       # addresses of the rows
            if x.dtype not in (torch.float16, torch.float32) or x.width % 4:
                return False
            k = x.width
        elif x.dim() != 2 or x.stride(1) != 1 or x.dtype not in (torch.float16, torch.float32) or x.requires_grad:
            return False
        else:
            k = x.size(1)
        for conv in model.convs:
            if conv.lin_l.bias is not None or conv.lin_l.weight.dtype != torch.float32 or k % 4:
                return False
            k = conv.lin_l.weight.size(0)
        return len(adjs) == len(model.convs)

    @staticmethod
    def forward(ctx, x, hops, training, p, *weights):
        L = nat.load()
        nat.require_device()
        n_layers = len(hops)
        st = _stream()
        n_id = refs = None
        if isinstance(x, tuple):                         # (table, n_id) of a TableRows: batch row j = table[n_id[j]]
            x, n_id = x
        elif isinstance(x, _Refs):                       # RowRefs: batch row j = the row at address addr[j]
            refs = x.r
            x = refs.addr
        h = x
        operands, acts, wcats, seeds = [], [], [], []
        for i, (rowptr, col, T) in enumerate(hops):
            K = refs.width if (i == 0 and refs is not None) else h.size(1)
            A = torch.empty((T, 2 * K), dtype=torch.float32, device=x.device)
            if i == 0 and refs is not None:
                nat.check(L.spp_sage_operand_forward_rows(_p(rowptr), _p(col), T, _p(refs.addr), int(refs.dtype == torch.float16),
                                                          K, _p(A), 2 * K, st))
            elif i == 0 and n_id is not None:
                nat.check(L.spp_sage_operand_forward_table(_p(rowptr), _p(col), T, _p(h), int(h.dtype == torch.float16),
                                                           h.stride(0) if h.size(0) > 1 else K, h.size(0), _p(n_id), K,
                                                           _p(A), 2 * K, st))
            elif i == 0:
                nat.check(L.spp_sage_operand_forward(_p(rowptr), _p(col), T, _p(h), int(h.dtype == torch.float16),
                                                     h.stride(0) if h.size(0) > 1 else K, K, _p(A), 2 * K, st))
            else:
                # h is the previous layer's PRE-activation: ReLU + dropout are applied to its rows as they are
                # loaded (no separate pass over the activation, which is never materialised)
                nat.check(L.spp_sage_operand_forward_act(_p(rowptr), _p(col), T, _p(h), K, _p(A), 2 * K, float(p),
                                                         int(bool(training)), seeds[i - 1], st))
            W = torch.cat([weights[2 * i], weights[2 * i + 1]], dim=1)          # [N, 2K] = [W_l | W_r]
            Z = _tall_linear(A, W)
            operands.append(A)
            wcats.append(W)
            if i != n_layers - 1:
                seeds.append(int(torch.empty((), dtype=torch.int64).random_().item()) if training else 0)
                acts.append(Z)                                                  # the pre-activation
                h = Z
            else:
                out = torch.log_softmax(Z, dim=-1)
        # tensors through save_for_backward (saved-tensor hooks, in-place version checks, a second backward with
        # retain_graph all behave as autograd users expect); only ints and seeds live on ctx
        hop_t = [t for (rowptr, col, _T) in hops for t in (rowptr, col)]
        ctx.save_for_backward(*operands, *acts, *wcats, out, *hop_t)
        ctx.hop_T = [int(T) for (_r, _c, T) in hops]
        ctx.act = (float(p), int(bool(training)), seeds)
        ctx.src_rows = [(x.numel() if refs is not None else x.size(0)) if n_id is None else n_id.numel()] + [a.size(0) for a in acts]
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_out):
        L = nat.load()
        st = _stream()
        n_layers = len(ctx.hop_T)
        sv = ctx.saved_tensors
        operands, acts = sv[:n_layers], sv[n_layers:2 * n_layers - 1]
        wcats, out = sv[2 * n_layers - 1:3 * n_layers - 1], sv[3 * n_layers - 1]
        hop_t = sv[3 * n_layers:]
        hops = [(hop_t[2 * i], hop_t[2 * i + 1], ctx.hop_T[i]) for i in range(n_layers)]
        gZ = torch._log_softmax_backward_data(g_out.contiguous(), out, -1, out.dtype)
        grads = [None] * (2 * n_layers)
        for i in range(n_layers - 1, -1, -1):
            A, W = operands[i], wcats[i]
            K = A.size(1) // 2
            gW = _wgrad(gZ, A)
            grads[2 * i], grads[2 * i + 1] = _split_wgrad(gW)
            if i == 0:
                break
            rowptr, col, T = hops[i]
            S = ctx.src_rows[i]
            gA = gZ @ W                                                         # [T, 2K]
            gH = torch.empty((S, K), dtype=torch.float32, device=gA.device)
            E = col.numel()
            p_, training_, seeds = ctx.act
            if E * K >= (1 << 22):
                # gather over the transposed hop, with the ReLU + dropout backward applied before the row is stored
                nbytes = int(L.spp_sage_operand_backward_workspace_bytes(T, S, E))
                ws = torch.empty(nbytes, dtype=torch.uint8, device=gA.device)
                nat.check(L.spp_sage_operand_backward_gather_act(_p(rowptr), _p(col), T, S, E, _p(gA), 2 * K, K, _p(gH),
                                                                 _p(ws), nbytes, _p(acts[i - 1]), p_, training_,
                                                                 seeds[i - 1], st))
            else:
                nat.check(L.spp_sage_operand_backward(_p(rowptr), _p(col), T, S, _p(gA), 2 * K, K, _p(gH), st))
                nat.check(L.spp_relu_dropout_backward_pre(_p(gH), _p(acts[i - 1]), gH.numel(), p_, training_,
                                                          seeds[i - 1], _p(gH), st))
            gZ = gH
        return (None, None, None, None, *grads)


# --------------------------------------------------------------------------------------------
# GAT  (driver/models.py:195-231: GATConv(bias=False, heads=1))
# --------------------------------------------------------------------------------------------
class _GatAggregate(torch.autograd.Function):
    """out[i] = sum_j softmax_j(leaky_relu(a_src[j] + a_dst[i])) h[j] over row i (its diagonal entry
    dropped) plus the self loop GATConv adds."""

    @staticmethod
    def forward(ctx, h, a_src, a_dst, rowptr, col, slope):
        L = nat.load()
        nat.require_device()
        h, a_src, a_dst = h.contiguous().float(), a_src.contiguous().float(), a_dst.contiguous().float()
        T, Fdim = a_dst.numel(), h.size(1)
        out = torch.empty((T, Fdim), dtype=torch.float32, device=h.device)
        rmax = torch.empty(T, dtype=torch.float32, device=h.device)
        rsum = torch.empty(T, dtype=torch.float32, device=h.device)
        nat.check(L.spp_gat_forward(_p(rowptr), _p(col), T, _p(h), Fdim, _p(a_src), _p(a_dst), float(slope), _p(out),
                                    _p(rmax), _p(rsum), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        ctx.save_for_backward(h, a_src, a_dst, rowptr, col, out, rmax, rsum)
        ctx.slope = float(slope)
        return out

    @staticmethod
    def backward(ctx, g):
        h, a_src, a_dst, rowptr, col, out, rmax, rsum = ctx.saved_tensors
        T, Fdim = a_dst.numel(), h.size(1)
        g = g.contiguous().float()
        grad_h = torch.zeros_like(h)
        grad_as = torch.zeros_like(a_src)
        grad_ad = torch.zeros_like(a_dst)
        nat.check(nat.load().spp_gat_backward(_p(rowptr), _p(col), T, _p(h), Fdim, _p(a_src), _p(a_dst), ctx.slope,
                                              _p(out), _p(rmax), _p(rsum), _p(g), _p(grad_h), _p(grad_as), _p(grad_ad),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return grad_h, grad_as, grad_ad, None, None, None


class GATConv(torch.nn.Module):
    """torch_geometric.nn.GATConv(in, out, heads=1, negative_slope=0.2, dropout=0, add_self_loops=True,
    bias=...) on a bipartite ((x, x_target), adj_t): one shared linear map for sources and targets
    (an int ``in_channels``), attention vectors att_src / att_dst, softmax over the incoming edges."""

    def __init__(self, in_channels, out_channels, heads=1, negative_slope=0.2, bias=True):
        super().__init__()
        if heads != 1:
            raise NotImplementedError("the reference model fixes heads=1 (driver/models.py:197)")
        self.negative_slope = negative_slope
        self.lin_src = torch.nn.Linear(in_channels, out_channels, bias=False)
        self.lin_dst = self.lin_src
        self.att_src = torch.nn.Parameter(torch.empty(1, 1, out_channels))
        self.att_dst = torch.nn.Parameter(torch.empty(1, 1, out_channels))
        self.bias = torch.nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        self.lin_src.reset_parameters()
        torch.nn.init.xavier_uniform_(self.att_src)          # PyG: glorot
        torch.nn.init.xavier_uniform_(self.att_dst)
        if self.bias is not None:
            torch.nn.init.zeros_(self.bias)

    def forward(self, x_pair, adj_t):
        x, x_target = x_pair
        rowptr, col, _ = adj_t.csr()
        T, K = x_target.size(0), x.size(1)
        if (x.is_cuda and x.dim() == 2 and x.stride(1) == 1 and x.dtype in (torch.float16, torch.float32) and
                K % 4 == 0 and K <= 1024 and x_target.data_ptr() == x.data_ptr() and x_target.stride() == x.stride()):
            # aggregate the raw rows with the attention weights, project only the T targets (_GatLayer)
            out = _GatLayer.apply(x, self.lin_src.weight, self.att_src.view(-1), self.att_dst.view(-1), rowptr, col, T,
                                  self.negative_slope)
            return out if self.bias is None else out + self.bias
        h = _TallLinear.apply(x.to(torch.float32), self.lin_src.weight)   # targets are the first rows of the sources
        h_t = h[:x_target.size(0)]
        a_src = (h * self.att_src.view(1, -1)).sum(-1)       # (a gemv on this tall shape measured 3x slower)
        a_dst = (h_t * self.att_dst.view(1, -1)).sum(-1)
        out = _GatAggregate.apply(h, a_src, a_dst, rowptr, col, self.negative_slope)
        return out if self.bias is None else out + self.bias


class _GatLayer(torch.autograd.Function):
    """GATConv(heads=1) on ((x, x[:T]), adj_t) as ONE node, in aggregate-then-project form:

        att . (W x_j) = x_j . (W^T att)          -> the logits need two K-vectors, not the projected rows
        sum_j alpha_ij (W x_j) = W sum_j alpha_ij x_j   -> aggregate raw rows, project the T targets only

    The same function as projecting all S source rows first (PyG's order), with S/T times less GEMM work
    and -- in the first layer -- 128-wide fp16 rows in the gather instead of 256-wide fp32 ones."""

    @staticmethod
    def forward(ctx, x, W, att_src, att_dst, rowptr, col, T, slope):
        L = nat.load()
        nat.require_device()
        st = _stream()
        S, K = x.size(0), x.size(1)
        half = int(x.dtype == torch.float16)
        xs = x.stride(0) if S > 1 else K
        v = torch.stack([att_src, att_dst]).to(torch.float32) @ W            # [2, K]: W^T att_src, W^T att_dst
        a_src = torch.empty(S, dtype=torch.float32, device=x.device)
        a_dst = torch.empty(T, dtype=torch.float32, device=x.device)
        nat.check(L.spp_gat_logits(_p(x), half, xs, S, T, K, _p(v[0]), _p(v[1]), _p(a_src), _p(a_dst), st))
        z = torch.empty((T, K), dtype=torch.float32, device=x.device)
        rmax = torch.empty(T, dtype=torch.float32, device=x.device)
        rsum = torch.empty(T, dtype=torch.float32, device=x.device)
        nat.check(L.spp_gat_aggregate_forward(_p(rowptr), _p(col), T, _p(x), half, xs, K, _p(a_src), _p(a_dst),
                                              float(slope), _p(z), _p(rmax), _p(rsum), st))
        ctx.save_for_backward(x, W, att_src, att_dst, rowptr, col, a_src, a_dst, z, rmax, rsum, v)
        ctx.dims = (S, T, K, half, xs, float(slope))
        return z @ W.t()

    @staticmethod
    def backward(ctx, g_out):
        L = nat.load()
        st = _stream()
        x, W, att_src, att_dst, rowptr, col, a_src, a_dst, z, rmax, rsum, v = ctx.saved_tensors
        S, T, K, half, xs, slope = ctx.dims
        g_out = g_out.contiguous()
        gW = _wgrad(g_out, z)                                                # [N, K]
        g_z = g_out @ W                                                      # [T, K]
        want_gx = ctx.needs_input_grad[0]
        E = col.numel()
        gather = want_gx and E * K >= (1 << 22) and K % 4 == 0
        g_as = torch.zeros(S, dtype=torch.float32, device=x.device)
        g_ad = torch.empty(T, dtype=torch.float32, device=x.device)
        if gather:
            # input gradient by gather over the transposed hop: no E x K fp32 atomics, no zero fill, and the two
            # rank-1 terms of the logits (a_src = x v_src, a_dst = x[:T] v_dst) are added in the same pass
            g_x = torch.empty((S, K), dtype=torch.float32, device=x.device)
            nbytes = int(L.spp_gat_aggregate_backward_gather_workspace_bytes(T, S, E))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            nat.check(L.spp_gat_aggregate_backward_gather(_p(rowptr), _p(col), T, S, E, _p(x), half, xs, K, _p(a_src),
                                                          _p(a_dst), slope, _p(z), _p(rmax), _p(rsum), _p(g_z), _p(v[0]),
                                                          _p(v[1]), _p(g_x), _p(g_as), _p(g_ad), _p(ws), nbytes, st))
        else:
            g_x = torch.zeros((S, K), dtype=torch.float32, device=x.device) if want_gx else None
            nat.check(L.spp_gat_aggregate_backward(_p(rowptr), _p(col), T, _p(x), half, xs, K, _p(a_src), _p(a_dst), slope,
                                                   _p(z), _p(rmax), _p(rsum), _p(g_z), _p(g_x), _p(g_as), _p(g_ad), st))
        g_v = torch.empty((2, K), dtype=torch.float32, device=x.device)
        nat.check(L.spp_gat_logits_backward(_p(x), half, xs, S, T, K, _p(g_as), _p(g_ad), _p(g_v[0]), _p(g_v[1]), st))
        if want_gx:                                                          # a_src = x v_src, a_dst = x[:T] v_dst
            if not gather:
                g_x.addr_(g_as, v[0])
                g_x[:T].addr_(g_ad, v[1])
            g_x = g_x.to(x.dtype)
        # v = [att_src; att_dst] @ W
        att = torch.stack([att_src, att_dst]).to(torch.float32)
        gW = gW + att.t() @ g_v                                              # [N, 2] @ [2, K]
        g_att = g_v @ W.t()                                                  # [2, N]
        return g_x, gW, g_att[0].to(att_src.dtype), g_att[1].to(att_dst.dtype), None, None, None, None


class GAT(torch.nn.Module):
    def __init__(self, in_channels, hidden_channels, out_channels, num_layers):
        super().__init__()
        self.num_layers = num_layers
        self.hidden_channels = hidden_channels
        self.convs = torch.nn.ModuleList()
        self.convs.append(GATConv(in_channels, hidden_channels, bias=False, heads=1))
        for _ in range(num_layers - 2):
            self.convs.append(GATConv(hidden_channels, hidden_channels, bias=False, heads=1))
        self.convs.append(GATConv(hidden_channels, out_channels, bias=False, heads=1))
        self.reset_parameters()

    def reset_parameters(self):
        for conv in self.convs:
            conv.reset_parameters()
            conv.apply(init_weights)

    def forward(self, x, adjs):
        if isinstance(x, (TableRows, RowRefs)):                     # (the fused first layer exists for SAGE)
            x = x.materialize()
        # the reference converts the features to fp32 first (models.py:221); GATConv here reads the fp16
        # rows directly (exact: every fp16 value is an fp32 value)
        for i, (adj_t, _e_id, size) in enumerate(adjs):
            x_target = x[:size[1]]
            x = self.convs[i]((x, x_target), adj_t)
            if i != self.num_layers - 1:
                x = relu_dropout(x, 0.5, self.training)
        return torch.log_softmax(x, dim=-1)
