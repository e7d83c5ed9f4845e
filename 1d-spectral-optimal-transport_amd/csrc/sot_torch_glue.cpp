// sot_torch_glue.cpp -- the host path of `Wasserstein1D(...)(x, y)` [+ `.backward()`] in C++ (round 3).
//
// Why: at the paper's own step size (64 clips x 16 frames = 1024 rows x 1025 bins, trainer.py:199-228) the kernels need ~10 us while
// the Python binding needed ~100 us of host time per forward + backward (ctypes marshalling of 15 arguments, three torch.empty, a
// Python autograd.Function whose backward re-enters the interpreter on the autograd engine's device thread).  This file is that
// host path as one pybind11 call: argument checks, output allocation through torch's caching allocator, ONE call into the C ABI
// (include/sot_hip.h -- the kernels stay behind it, this file contains no device code and no arithmetic on tensor data) and a
// C++ torch::autograd::Function whose backward rescales the gradient that the training-form kernel produced together with the
// loss (sot_w1d_loss_and_grad), without taking the GIL.
//
// Covers the module's hot call: 2-D float32 HIP tensors, shared (planned) positions, default reduction (mean over every row), no
// hinge, gradient wanted for y only or not at all.  Everything else stays on the Python binding (_native.py), which remains the
// complete implementation.  The library is the process's libsot_hip.so (bound by path with dlopen: the same handle ctypes holds).
#include <torch/extension.h>
#include <torch/csrc/autograd/custom_function.h>
// PyTorch-ROCm presents HIP devices as DeviceType::CUDA ("masquerading"): the guard and the stream accessor of that convention
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <dlfcn.h>
#include <string>

#include "../../include/sot_hip.h"

namespace {

struct Api {
    decltype(&sot_abi_version) abi = nullptr;
    decltype(&sot_status_string) status_string = nullptr;
    decltype(&sot_w1d_loss) loss = nullptr;
    decltype(&sot_w1d_loss_and_grad) loss_and_grad = nullptr;
    decltype(&sot_w1d_backward) backward = nullptr;
    decltype(&sot_scale_inplace) scale_inplace = nullptr;
    decltype(&sot_prepare_positions) prepare_positions = nullptr;
    decltype(&sot_prepare_unit_positions) prepare_unit_positions = nullptr;
    decltype(&sot_stft_frames) stft_frames = nullptr;
    decltype(&sot_stft_mag_forward_pair_spec) stft_pair = nullptr;
    decltype(&sot_stft_mag_backward_spec) stft_backward = nullptr;
    decltype(&sot_stft_backward_workspace_bytes) stft_backward_ws = nullptr;
    decltype(&sot_stft_mag_forward_spec) stft_forward = nullptr;
    decltype(&sot_mss_workspace_bytes) mss_ws = nullptr;
    decltype(&sot_mss_loss_and_grad) mss = nullptr;
};
Api g_api;

template <typename F>
void bind_symbol(void* handle, const char* name, F& slot)
{
    void* sym = dlsym(handle, name);
    TORCH_CHECK(sym != nullptr, "sot glue: libsot_hip.so does not export ", name);
    slot = reinterpret_cast<F>(sym);
}

int64_t bind_library(const std::string& path)
{
    void* handle = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    TORCH_CHECK(handle != nullptr, "sot glue: cannot open ", path, ": ", dlerror());
    bind_symbol(handle, "sot_abi_version", g_api.abi);
    bind_symbol(handle, "sot_status_string", g_api.status_string);
    bind_symbol(handle, "sot_w1d_loss", g_api.loss);
    bind_symbol(handle, "sot_w1d_loss_and_grad", g_api.loss_and_grad);
    bind_symbol(handle, "sot_w1d_backward", g_api.backward);
    bind_symbol(handle, "sot_scale_inplace", g_api.scale_inplace);
    bind_symbol(handle, "sot_prepare_positions", g_api.prepare_positions);
    bind_symbol(handle, "sot_prepare_unit_positions", g_api.prepare_unit_positions);
    bind_symbol(handle, "sot_stft_frames", g_api.stft_frames);
    bind_symbol(handle, "sot_stft_mag_forward_pair_spec", g_api.stft_pair);
    bind_symbol(handle, "sot_stft_mag_backward_spec", g_api.stft_backward);
    bind_symbol(handle, "sot_stft_backward_workspace_bytes", g_api.stft_backward_ws);
    bind_symbol(handle, "sot_stft_mag_forward_spec", g_api.stft_forward);
    bind_symbol(handle, "sot_mss_workspace_bytes", g_api.mss_ws);
    bind_symbol(handle, "sot_mss_loss_and_grad", g_api.mss);
    TORCH_CHECK(g_api.abi() == SOT_ABI_VERSION, "sot glue: libsot_hip.so has ABI version ", g_api.abi(), ", this glue was built for ",
                SOT_ABI_VERSION);
    return g_api.abi();
}

void check_status(int rc, double p)
{
    if (rc == SOT_OK) return;
    // same exception type and text as losses.py:271
    if (rc == SOT_ERR_INVALID_P) {
        PyErr_SetString(PyExc_AssertionError, ("The OT loss is only valid for p>=1, " + std::to_string(p) + " was given").c_str());
        throw pybind11::error_already_set();
    }
    TORCH_CHECK(false, "libsot_hip: ", g_api.status_string(rc), " (status ", rc, ")");
}

// the module's rows: 2-D float32 HIP tensors with unit inner stride
void check_rows(const at::Tensor& t, const char* what)
{
    TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kFloat && t.dim() == 2, "sot glue: ", what, " must be a 2-D float32 GPU tensor");
    TORCH_CHECK(t.size(1) >= 1 && (t.size(1) == 1 || t.stride(1) == 1) && (t.size(0) <= 1 || t.stride(0) >= t.size(1)),
                "sot glue: ", what, " must have unit inner stride and non-overlapping rows");
}

struct Plan {   // the position plan of sot_prepare_positions (sorted positions, permutations, identity flags), kept by the Python side
    at::Tensor xs, ys, xperm, yperm, ident;
};

sot_problem make_problem(const at::Tensor& x, const at::Tensor& y, const Plan& plan, double p, int64_t flags)
{
    check_rows(x, "x");
    check_rows(y, "y");
    TORCH_CHECK(x.size(0) == y.size(0), "row count mismatch: ", x.size(0), " vs ", y.size(0));
    TORCH_CHECK(plan.xs.numel() == x.size(1) && plan.ys.numel() == y.size(1), "positions and weights must have the same number of features");
    TORCH_CHECK(x.device() == y.device() && plan.xs.device() == x.device(), "sot glue: all tensors must live on one GPU");
    sot_problem pr{};
    pr.x = x.data_ptr<float>();
    pr.y = y.data_ptr<float>();
    pr.xpos = plan.xs.data_ptr<float>();
    pr.ypos = plan.ys.data_ptr<float>();
    pr.B = x.size(0);
    pr.n = (int32_t)x.size(1);
    pr.m = (int32_t)y.size(1);
    pr.x_row_stride = pr.B > 1 ? x.stride(0) : pr.n;
    pr.y_row_stride = pr.B > 1 ? y.stride(0) : pr.m;
    pr.xpos_row_stride = pr.ypos_row_stride = 0;
    pr.p = (float)p;
    pr.flags = (uint32_t)flags;
    pr.xperm = plan.xperm.data_ptr<int32_t>();
    pr.yperm = plan.yperm.data_ptr<int32_t>();
    pr.perm_is_identity = plan.ident.data_ptr<int32_t>();
    return pr;
}

void* current_stream(const at::Tensor& t) { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }

// forward + batch mean, no gradient (sot_w1d_loss): the eval / no-grad call of the module
at::Tensor mean_loss_nograd(const at::Tensor& x, const at::Tensor& y, const Plan& plan, double p, int64_t flags)
{
    sot_problem pr = make_problem(x, y, plan, p, flags);
    TORCH_CHECK(pr.B > 0, "libsot_hip: bad shape or stride (status ", (int)SOT_ERR_BAD_SHAPE, ")");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    at::Tensor rows = at::empty({pr.B}, x.options());
    at::Tensor mean = at::empty({}, x.options());
    check_status(g_api.loss(&pr, rows.data_ptr<float>(), (double)pr.B, 0, 0.0f, mean.data_ptr<float>(), nullptr, nullptr, nullptr, 0,
                            current_stream(x)), p);
    return mean;
}

// The training step's node: the loss AND d mean / d y out of one pass over the rows (sot_w1d_loss_and_grad); the backward multiplies
// the stored gradient by the upstream scalar (a kernel that exits at once when that scalar is 1).  A second backward through a
// retained graph recomputes the gradient with the backward kernel (sot_w1d_backward).
class FusedMeanLoss : public torch::autograd::Function<FusedMeanLoss> {
public:
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& x, const at::Tensor& y, const at::Tensor& xs,
                              const at::Tensor& ys, const at::Tensor& xperm, const at::Tensor& yperm, const at::Tensor& ident, double p,
                              int64_t flags)
    {
        const Plan plan{xs, ys, xperm, yperm, ident};
        sot_problem pr = make_problem(x, y, plan, p, flags);
        TORCH_CHECK(pr.B > 0, "libsot_hip: bad shape or stride (status ", (int)SOT_ERR_BAD_SHAPE, ")");
        const c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
        at::Tensor rows = at::empty({pr.B}, x.options());
        at::Tensor mean = at::empty({}, x.options());
        at::Tensor gy = at::empty({pr.B, (int64_t)pr.m}, x.options());
        check_status(g_api.loss_and_grad(&pr, rows.data_ptr<float>(), (double)pr.B, mean.data_ptr<float>(), nullptr, (float)(1.0 / (double)pr.B),
                                         gy.data_ptr<float>(), nullptr, nullptr, 0, current_stream(x)), p);
        ctx->save_for_backward({x, y, xs, ys, xperm, yperm, ident});
        ctx->saved_data["gy"] = gy;
        ctx->saved_data["p"] = p;
        ctx->saved_data["flags"] = flags;
        return mean;
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grad_outputs)
    {
        at::Tensor g = grad_outputs[0];
        if (g.scalar_type() != at::kFloat) g = g.to(at::kFloat);
        g = g.contiguous();
        const c10::hip::HIPGuardMasqueradingAsCUDA guard(g.device());
        at::Tensor gy;
        const c10::IValue held = ctx->saved_data["gy"];
        if (held.isTensor() && held.toTensor().defined()) {
            gy = held.toTensor();
            ctx->saved_data["gy"] = c10::IValue();   // consumed once
            const int rc = g_api.scale_inplace(gy.data_ptr<float>(), gy.numel(), g.data_ptr<float>(), current_stream(gy));
            check_status(rc, 1.0);
        } else {   // a second backward through a retained graph
            const auto saved = ctx->get_saved_variables();
            const double p = ctx->saved_data["p"].toDouble();
            const Plan plan{saved[2], saved[3], saved[4], saved[5], saved[6]};
            sot_problem pr = make_problem(saved[0], saved[1], plan, p, ctx->saved_data["flags"].toInt());
            gy = at::empty({pr.B, (int64_t)pr.m}, saved[1].options());
            const int rc = g_api.backward(&pr, g.data_ptr<float>(), 0, (float)(1.0 / (double)pr.B), nullptr, gy.data_ptr<float>(), nullptr, 0,
                                          current_stream(gy));
            check_status(rc, p);
        }
        return {at::Tensor(), gy, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

// The module's default call.  Gradient wanted for y only -> the autograd node above; none -> the forward alone.  (x or the positions
// wanting a gradient is the caller's case to route elsewhere: the Python binding has those paths.)
at::Tensor mean_loss(const at::Tensor& x, const at::Tensor& y, const at::Tensor& xs, const at::Tensor& ys, const at::Tensor& xperm,
                     const at::Tensor& yperm, const at::Tensor& ident, double p, int64_t flags)
{
    TORCH_CHECK(g_api.loss != nullptr, "sot glue: bind() has not been called");
    const bool grad = at::GradMode::is_enabled() && y.requires_grad();
    TORCH_CHECK(!(at::GradMode::is_enabled() && x.requires_grad()), "sot glue: a gradient w.r.t. x is not this path's case");
    if (grad) return FusedMeanLoss::apply(x, y, xs, ys, xperm, yperm, ident, p, flags);
    return mean_loss_nograd(x, y, Plan{xs, ys, xperm, yperm, ident}, p, flags);
}

// The position plan of one pair of shared grids (sot_prepare_positions) with all five outputs in ONE allocation: the reference's
// trainer builds x_pos / y_pos afresh every step (trainer.py:187-197), so a plan per step must cost one allocation and one launch,
// not five allocations.  Returns (sorted x positions, sorted y positions, x permutation, y permutation, identity flags): views of one
// buffer, each 256-byte aligned.
std::vector<at::Tensor> make_plan(const at::Tensor& xpos, const at::Tensor& ypos, bool unit = false)
{
    TORCH_CHECK(g_api.prepare_positions != nullptr, "sot glue: bind() has not been called");
    TORCH_CHECK(xpos.is_cuda() && ypos.is_cuda() && xpos.scalar_type() == at::kFloat && ypos.scalar_type() == at::kFloat && xpos.dim() == 1 &&
                ypos.dim() == 1 && xpos.is_contiguous() && ypos.is_contiguous() && xpos.numel() >= 1 && ypos.numel() >= 1,
                "sot glue: positions must be contiguous 1-D float32 GPU tensors");
    const int64_t n = xpos.numel(), m = ypos.numel();
    auto pad = [](int64_t v) { return (v + 63) / 64 * 64; };   // in 4-byte words: 256-byte aligned pieces
    const int64_t o_ys = pad(n), o_xp = o_ys + pad(m), o_yp = o_xp + pad(n), o_id = o_yp + pad(m), total = o_id + 64;
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(xpos.device());
    at::Tensor buf = at::empty({total}, xpos.options());
    at::Tensor ints = buf.view(at::kInt);
    at::Tensor xs = buf.narrow(0, 0, n), ys = buf.narrow(0, o_ys, m);
    at::Tensor xperm = ints.narrow(0, o_xp, n), yperm = ints.narrow(0, o_yp, m), ident = ints.narrow(0, o_id, 2);
    // unit: both grids divided by their maxima inside the launch (sot_prepare_unit_positions: trainer.py:196-197 without the three torch kernels)
    check_status((unit ? g_api.prepare_unit_positions : g_api.prepare_positions)(
                     xpos.data_ptr<float>(), ypos.data_ptr<float>(), (int32_t)n, (int32_t)m, xs.data_ptr<float>(), ys.data_ptr<float>(),
                     xperm.data_ptr<int32_t>(), yperm.data_ptr<int32_t>(), ident.data_ptr<int32_t>(), current_stream(xpos)), 1.0);
    return {xs, ys, xperm, yperm, ident};
}

// The module's default call with FRESH position tensors (round 4).  The reference's trainer rebuilds its grid on every step
// (trainer.py:187-197: x_pos = torch.tensor(freqs).to(device) / max, y_pos = x_pos.clone()), so identity-keyed plan and hot-call caches
// never hit there.  The grid's CONTENT cannot be compared on the host without a synchronisation, and a cached plan's buffers must not be
// rewritten in place (an autograd node of an earlier step may still hold them for a retained graph), so each such call gets its own
// plan -- but as part of THIS one host call: one allocation, sot_prepare_positions (two workgroups: a sortedness check, and a sort
// only when it fails), then the loss.  The flag word must not carry SOT_FLAG_SAME_GRID unless both arguments are the same tensor:
// whether two fresh tensors hold one grid is device-side knowledge.
at::Tensor mean_loss_fresh(const at::Tensor& x, const at::Tensor& y, const at::Tensor& xpos, const at::Tensor& ypos, double p, int64_t flags)
{
    TORCH_CHECK(!(at::GradMode::is_enabled() && (xpos.requires_grad() || ypos.requires_grad())),
                "sot glue: gradients w.r.t. the positions are not this path's case");
    TORCH_CHECK(xpos.device() == x.device() && ypos.device() == x.device(), "sot glue: all tensors must live on one GPU");
    if (!(xpos.is_same(ypos) || (xpos.data_ptr() == ypos.data_ptr() && xpos.numel() == ypos.numel()))) flags &= ~(int64_t)SOT_FLAG_SAME_GRID;
    const std::vector<at::Tensor> plan = make_plan(xpos, ypos);
    return mean_loss(x, y, plan[0], plan[1], plan[2], plan[3], plan[4], p, flags);
}

// ---- the training-step slice trainer.py:192-228 runs around the loss, audio in (spectra.training_step_slice / _AudioToLoss): magnitude
// STFT of target and estimate in one launch (keeping the estimate's complex spectrum), SOT loss + d mean / d spectrum in one pass,
// and on the way back the STFT backward from the stored spectrum with the upstream scalar applied inside it.  Differentiates w.r.t.
// the estimate's audio only.
class AudioToLoss : public torch::autograd::Function<AudioToLoss> {
public:
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& target, const at::Tensor& estimate,
                              const at::Tensor& window, const at::Tensor& xs, const at::Tensor& ys, const at::Tensor& xperm,
                              const at::Tensor& yperm, const at::Tensor& ident, int64_t n_fft, int64_t hop, double p, int64_t flags, bool grad)
    {
        TORCH_CHECK(target.is_cuda() && target.scalar_type() == at::kFloat && target.dim() == 2 && target.is_contiguous() &&
                    estimate.is_cuda() && estimate.scalar_type() == at::kFloat && estimate.is_contiguous() && estimate.sizes() == target.sizes(),
                    "sot glue: target and estimate must be contiguous float32 GPU tensors [clips, samples] of one shape");
        TORCH_CHECK(window.is_cuda() && window.scalar_type() == at::kFloat && window.is_contiguous() && window.numel() == n_fft &&
                    reinterpret_cast<uintptr_t>(window.data_ptr<float>()) % 8 == 0, "sot glue: window must hold n_fft float32 taps, 8-byte aligned");
        const int64_t clips = target.size(0), samples = target.size(1);
        const int64_t frames = g_api.stft_frames(samples, (int)hop), bins = n_fft / 2 + 1;
        const c10::hip::HIPGuardMasqueradingAsCUDA guard(target.device());
        void* st = current_stream(target);
        at::Tensor mag = at::empty({2 * clips, frames, bins}, target.options());
        at::Tensor cplx = grad ? at::empty({clips, frames, bins, 2}, target.options()) : at::Tensor();
        check_status(g_api.stft_pair(target.data_ptr<float>(), samples, estimate.data_ptr<float>(), samples, clips, samples,
                                     window.data_ptr<float>(), (int)n_fft, (int)hop, mag.data_ptr<float>(),
                                     grad ? cplx.data_ptr<float>() : nullptr, st), p);
        const at::Tensor rows_x = mag.narrow(0, 0, clips).view({clips * frames, bins});
        const at::Tensor rows_y = mag.narrow(0, clips, clips).view({clips * frames, bins});
        const Plan plan{xs, ys, xperm, yperm, ident};
        sot_problem pr = make_problem(rows_x, rows_y, plan, p, flags);
        at::Tensor rows = at::empty({pr.B}, target.options());
        at::Tensor mean = at::empty({}, target.options());
        if (grad) {
            at::Tensor gy = at::empty({clips, frames, bins}, target.options());
            check_status(g_api.loss_and_grad(&pr, rows.data_ptr<float>(), (double)pr.B, mean.data_ptr<float>(), nullptr, (float)(1.0 / (double)pr.B),
                                             gy.data_ptr<float>(), nullptr, nullptr, 0, st), p);
            ctx->save_for_backward({window});
            ctx->saved_data["gy"] = gy;
            ctx->saved_data["cplx"] = cplx;
            ctx->saved_data["n_fft"] = n_fft;
            ctx->saved_data["hop"] = hop;
            ctx->saved_data["samples"] = samples;
        } else {
            check_status(g_api.loss(&pr, rows.data_ptr<float>(), (double)pr.B, 0, 0.0f, mean.data_ptr<float>(), nullptr, nullptr, nullptr, 0, st), p);
        }
        return mean;
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grad_outputs)
    {
        at::Tensor g = grad_outputs[0];
        if (g.scalar_type() != at::kFloat) g = g.to(at::kFloat);
        g = g.contiguous();
        const at::Tensor gy = ctx->saved_data["gy"].toTensor(), cplx = ctx->saved_data["cplx"].toTensor();
        const at::Tensor window = ctx->get_saved_variables()[0];
        const int64_t n_fft = ctx->saved_data["n_fft"].toInt(), hop = ctx->saved_data["hop"].toInt(), samples = ctx->saved_data["samples"].toInt();
        const int64_t clips = gy.size(0);
        const c10::hip::HIPGuardMasqueradingAsCUDA guard(gy.device());
        at::Tensor grad_audio = at::empty({clips, samples}, gy.options());
        const size_t ws_bytes = g_api.stft_backward_ws(clips, samples, (int)n_fft, (int)hop);
        at::Tensor ws = at::empty({(int64_t)(ws_bytes > 0 ? ws_bytes : 1)}, gy.options().dtype(at::kByte));
        // the gradient is read, never modified: a second backward through a retained graph runs the same call again
        check_status(g_api.stft_backward(nullptr, cplx.data_ptr<float>(), clips, samples, samples, window.data_ptr<float>(), (int)n_fft, (int)hop,
                                         gy.data_ptr<float>(), g.data_ptr<float>(), grad_audio.data_ptr<float>(), 0, ws.data_ptr(), ws_bytes,
                                         current_stream(gy)), 1.0);
        return {at::Tensor(), grad_audio, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(),
                at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

at::Tensor audio_to_loss(const at::Tensor& target, const at::Tensor& estimate, const at::Tensor& window, const at::Tensor& xs,
                         const at::Tensor& ys, const at::Tensor& xperm, const at::Tensor& yperm, const at::Tensor& ident, int64_t n_fft,
                         int64_t hop, double p, int64_t flags)
{
    TORCH_CHECK(g_api.stft_pair != nullptr, "sot glue: bind() has not been called");
    TORCH_CHECK(!(at::GradMode::is_enabled() && target.requires_grad()), "sot glue: a gradient w.r.t. the target is not this path's case");
    const bool grad = at::GradMode::is_enabled() && estimate.requires_grad();   // decided here: forward() runs with grad mode off
    return AudioToLoss::apply(target, estimate, window, xs, ys, xperm, yperm, ident, n_fft, hop, p, flags, grad);
}

// ---- round 5: the two other modules of the trainer's loss block (trainer.py:199-221) on the same kind of host path: the transform
// (`features.TorchSTFT.forward`: spectra.stft_magnitude) and `MSSLoss`.  Launched from Python through ctypes + a Python autograd.Function each
// costs 40-60 us of host time per call; the paper's step makes five such calls.

// [clips, samples] audio -> [clips, frames, n_fft / 2 + 1] magnitudes; keeps the complex spectrum for the backward when the audio is differentiated
class StftMagnitude : public torch::autograd::Function<StftMagnitude> {
public:
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& audio, const at::Tensor& window, int64_t n_fft, int64_t hop, bool grad)
    {
        TORCH_CHECK(audio.is_cuda() && audio.scalar_type() == at::kFloat && audio.dim() == 2 && audio.is_contiguous(),
                    "sot glue: audio must be a contiguous float32 GPU tensor [clips, samples]");
        TORCH_CHECK(window.is_cuda() && window.scalar_type() == at::kFloat && window.is_contiguous() && window.numel() == n_fft &&
                    reinterpret_cast<uintptr_t>(window.data_ptr<float>()) % 8 == 0, "sot glue: window must hold n_fft float32 taps, 8-byte aligned");
        const int64_t clips = audio.size(0), samples = audio.size(1);
        const int64_t frames = g_api.stft_frames(samples, (int)hop), bins = n_fft / 2 + 1;
        const c10::hip::HIPGuardMasqueradingAsCUDA guard(audio.device());
        at::Tensor mag = at::empty({clips, frames, bins}, audio.options());
        at::Tensor cplx = grad ? at::empty({clips, frames, bins, 2}, audio.options()) : at::Tensor();
        check_status(g_api.stft_forward(audio.data_ptr<float>(), clips, samples, samples, window.data_ptr<float>(), (int)n_fft, (int)hop,
                                        mag.data_ptr<float>(), grad ? cplx.data_ptr<float>() : nullptr, current_stream(audio)), 1.0);
        if (grad) {
            ctx->save_for_backward({window});
            ctx->saved_data["cplx"] = cplx;
            ctx->saved_data["n_fft"] = n_fft;
            ctx->saved_data["hop"] = hop;
            ctx->saved_data["samples"] = samples;
        }
        return mag;
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grad_outputs)
    {
        at::Tensor g = grad_outputs[0];
        if (g.scalar_type() != at::kFloat) g = g.to(at::kFloat);
        g = g.contiguous();
        const at::Tensor cplx = ctx->saved_data["cplx"].toTensor();
        const at::Tensor window = ctx->get_saved_variables()[0];
        const int64_t n_fft = ctx->saved_data["n_fft"].toInt(), hop = ctx->saved_data["hop"].toInt(), samples = ctx->saved_data["samples"].toInt();
        const int64_t clips = cplx.size(0);
        const c10::hip::HIPGuardMasqueradingAsCUDA guard(g.device());
        at::Tensor grad_audio = at::empty({clips, samples}, g.options());
        const size_t ws_bytes = g_api.stft_backward_ws(clips, samples, (int)n_fft, (int)hop);
        at::Tensor ws = at::empty({(int64_t)(ws_bytes > 0 ? ws_bytes : 1)}, g.options().dtype(at::kByte));
        check_status(g_api.stft_backward(nullptr, cplx.data_ptr<float>(), clips, samples, samples, window.data_ptr<float>(), (int)n_fft, (int)hop,
                                         g.data_ptr<float>(), nullptr, grad_audio.data_ptr<float>(), 0, ws.data_ptr(), ws_bytes, current_stream(g)), 1.0);
        return {grad_audio, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

at::Tensor stft_magnitude(const at::Tensor& audio, const at::Tensor& window, int64_t n_fft, int64_t hop)
{
    TORCH_CHECK(g_api.stft_forward != nullptr, "sot glue: bind() has not been called");
    const bool grad = at::GradMode::is_enabled() && audio.requires_grad();   // decided here: forward() runs with grad mode off
    return StftMagnitude::apply(audio, window, n_fft, hop, grad);
}

// MSSLoss in two launches (sot_mss_loss_and_grad): the loss and d loss / d estimate from the forward; the backward multiplies by the upstream
// gradient (one scalar, or one value per clip for the `dims` = (1, 2) form)
class MssLoss : public torch::autograd::Function<MssLoss> {
public:
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& target, const at::Tensor& estimate,
                              const std::vector<at::Tensor>& windows, const std::vector<int64_t>& fft_sizes, double mag_weight, double logmag_weight,
                              bool l2, bool per_clip, bool grad)
    {
        TORCH_CHECK(target.is_cuda() && target.scalar_type() == at::kFloat && target.dim() == 2 && estimate.is_cuda() &&
                    estimate.scalar_type() == at::kFloat && estimate.sizes() == target.sizes() && target.stride(1) == 1 && estimate.stride(1) == 1,
                    "sot glue: target and estimate must be float32 GPU tensors [clips, samples] of one shape with unit inner stride");
        const int n = (int)fft_sizes.size();
        TORCH_CHECK(n >= 1 && n <= 8 && (int)windows.size() == n, "sot glue: one window per FFT size, at most 8");
        int sizes[8];
        const float* wins[8];
        for (int i = 0; i < n; ++i) {
            sizes[i] = (int)fft_sizes[i];
            TORCH_CHECK(windows[i].is_cuda() && windows[i].scalar_type() == at::kFloat && windows[i].is_contiguous() && windows[i].numel() == fft_sizes[i] &&
                        reinterpret_cast<uintptr_t>(windows[i].data_ptr<float>()) % 8 == 0, "sot glue: window ", i, " must hold n_fft float32 taps, 8-byte aligned");
            wins[i] = windows[i].data_ptr<float>();
        }
        const int64_t clips = target.size(0), samples = target.size(1);
        const c10::hip::HIPGuardMasqueradingAsCUDA guard(target.device());
        const size_t ws_bytes = g_api.mss_ws(clips, samples, sizes, n);
        TORCH_CHECK(ws_bytes > 0 || clips == 0, "libsot_hip: unsupported size (status ", (int)SOT_ERR_UNSUPPORTED_SIZE, ")");
        at::Tensor ws = at::empty({(int64_t)(ws_bytes > 0 ? ws_bytes : 8)}, target.options().dtype(at::kByte));
        at::Tensor loss = per_clip ? at::empty({clips}, target.options()) : at::empty({}, target.options());
        at::Tensor gv = grad ? at::empty({clips, samples}, target.options()) : at::Tensor();
        check_status(g_api.mss(target.data_ptr<float>(), clips > 1 ? target.stride(0) : samples, estimate.data_ptr<float>(),
                               clips > 1 ? estimate.stride(0) : samples, clips, samples, sizes, wins, n, (float)mag_weight, (float)logmag_weight, 1e-5f,
                               l2 ? 1 : 0, per_clip ? 1 : 0, 1.0f, loss.data_ptr<float>(), grad ? gv.data_ptr<float>() : nullptr, ws.data_ptr(), ws_bytes,
                               current_stream(target)), 1.0);
        if (grad) ctx->saved_data["gv"] = gv;
        ctx->saved_data["per_clip"] = per_clip;
        return loss;
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grad_outputs)
    {
        at::Tensor g = grad_outputs[0];
        if (g.scalar_type() != at::kFloat) g = g.to(at::kFloat);
        const at::Tensor gv = ctx->saved_data["gv"].toTensor();
        const c10::hip::HIPGuardMasqueradingAsCUDA guard(gv.device());
        at::Tensor out = ctx->saved_data["per_clip"].toBool() ? gv * g.reshape({-1, 1}) : gv * g;   // the stored gradient is never modified
        return {at::Tensor(), out, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

at::Tensor mss_loss(const at::Tensor& target, const at::Tensor& estimate, const std::vector<at::Tensor>& windows, const std::vector<int64_t>& fft_sizes,
                    double mag_weight, double logmag_weight, bool l2, bool per_clip)
{
    TORCH_CHECK(g_api.mss != nullptr, "sot glue: bind() has not been called");
    TORCH_CHECK(!(at::GradMode::is_enabled() && target.requires_grad()), "sot glue: a gradient w.r.t. the target is not this path's case");
    const bool grad = at::GradMode::is_enabled() && estimate.requires_grad();
    return MssLoss::apply(target, estimate, windows, fft_sizes, mag_weight, logmag_weight, l2, per_clip, grad);
}

// ---- round 6: the WHOLE loss block of the paper's training step (trainer.py:183-245 with train_config.yaml:73-102: `MixOfLosses([MSSLoss,
// Wasserstein1D], weights)` on the audio pair and on its STFT magnitudes, the sum of the two means, gradient into the estimate's audio) as ONE
// host call and ONE autograd node.  Module by module the step is ~25 launches, fifteen of them the trainer's own arithmetic (`* weight`,
// `.mean()` of a scalar, `0 + value`, the autograd engine's gradient accumulation): ~44 of its 127 us at the paper's 64 clips.  Here the
// forward runs plan -> STFT pair -> SOT loss + d mean / d spectrum -> STFT backward from the stored spectrum, accumulated INTO the gradient
// the MSS kernels wrote (their weights carry the mix weight; the SOT's goes into its gradient scale), and one addition of the two scalars.  The
// backward multiplies the stored gradient by the upstream scalar.  (The MSS kernels on a second stream beside the STFT -> SOT chain -- the
// two are independent until the accumulation -- measured no gain: 96.0 against 97.1 us replayed at 64 clips, and 50 us more host time eager.)
class MixLossStep : public torch::autograd::Function<MixLossStep> {
public:
    static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, const at::Tensor& target, const at::Tensor& estimate,
                              const at::Tensor& window, const at::Tensor& xpos, const at::Tensor& ypos, int64_t n_fft, int64_t hop, double p, int64_t flags,
                              const std::vector<at::Tensor>& mss_windows, const std::vector<int64_t>& mss_sizes, double mag_weight,
                              double logmag_weight, bool l2, double w_mss, double w_sot, bool unit_positions, bool grad)
    {
        TORCH_CHECK(target.is_cuda() && target.scalar_type() == at::kFloat && target.dim() == 2 && target.is_contiguous() &&
                    estimate.is_cuda() && estimate.scalar_type() == at::kFloat && estimate.is_contiguous() && estimate.sizes() == target.sizes() &&
                    target.size(0) > 0, "sot glue: target and estimate must be contiguous float32 GPU tensors [clips, samples] of one shape");
        TORCH_CHECK(window.is_cuda() && window.scalar_type() == at::kFloat && window.is_contiguous() && window.numel() == n_fft &&
                    reinterpret_cast<uintptr_t>(window.data_ptr<float>()) % 8 == 0, "sot glue: window must hold n_fft float32 taps, 8-byte aligned");
        const int n = (int)mss_sizes.size();
        TORCH_CHECK(n >= 1 && n <= 8 && (int)mss_windows.size() == n, "sot glue: one window per FFT size, at most 8");
        int sizes[8];
        const float* wins[8];
        for (int i = 0; i < n; ++i) {
            sizes[i] = (int)mss_sizes[i];
            TORCH_CHECK(mss_windows[i].is_cuda() && mss_windows[i].scalar_type() == at::kFloat && mss_windows[i].is_contiguous() &&
                        mss_windows[i].numel() == mss_sizes[i] && reinterpret_cast<uintptr_t>(mss_windows[i].data_ptr<float>()) % 8 == 0,
                        "sot glue: window ", i, " must hold n_fft float32 taps, 8-byte aligned");
            wins[i] = mss_windows[i].data_ptr<float>();
        }
        const int64_t clips = target.size(0), samples = target.size(1);
        const int64_t frames = g_api.stft_frames(samples, (int)hop), bins = n_fft / 2 + 1;
        const c10::hip::HIPGuardMasqueradingAsCUDA guard(target.device());
        void* st = current_stream(target);
        const size_t mss_bytes = g_api.mss_ws(clips, samples, sizes, n);
        TORCH_CHECK(mss_bytes > 0, "libsot_hip: unsupported size (status ", (int)SOT_ERR_UNSUPPORTED_SIZE, ")");
        at::Tensor mss_ws = at::empty({(int64_t)mss_bytes}, target.options().dtype(at::kByte));
        at::Tensor mss_loss = at::empty({}, target.options());
        at::Tensor grad_audio = grad ? at::empty({clips, samples}, target.options()) : at::Tensor();
        at::Tensor mag = at::empty({2 * clips, frames, bins}, target.options());
        at::Tensor cplx = grad ? at::empty({clips, frames, bins, 2}, target.options()) : at::Tensor();

        // fresh plan, both magnitude spectrograms in one launch, SOT loss (+ d (w_sot * mean) / d spectrum)
        if (!(xpos.is_same(ypos) || (xpos.data_ptr() == ypos.data_ptr() && xpos.numel() == ypos.numel()))) flags &= ~(int64_t)SOT_FLAG_SAME_GRID;
        // unit_positions: xpos / ypos are the transform's bin frequencies as they are; x_pos = f / f.max() and y_pos = x_pos.clone() happen inside
        // the plan's launch -- two separate grids to the kernels, as in the trainer (never SOT_FLAG_SAME_GRID)
        if (unit_positions) flags &= ~(int64_t)SOT_FLAG_SAME_GRID;
        const std::vector<at::Tensor> pl = make_plan(xpos, ypos, unit_positions);
        check_status(g_api.stft_pair(target.data_ptr<float>(), samples, estimate.data_ptr<float>(), samples, clips, samples,
                                     window.data_ptr<float>(), (int)n_fft, (int)hop, mag.data_ptr<float>(),
                                     grad ? cplx.data_ptr<float>() : nullptr, st), p);
        const at::Tensor rows_x = mag.narrow(0, 0, clips).view({clips * frames, bins});
        const at::Tensor rows_y = mag.narrow(0, clips, clips).view({clips * frames, bins});
        const Plan plan{pl[0], pl[1], pl[2], pl[3], pl[4]};
        sot_problem pr = make_problem(rows_x, rows_y, plan, p, flags);
        at::Tensor rows = at::empty({pr.B}, target.options());
        at::Tensor sot_mean = at::empty({}, target.options());
        at::Tensor gy;
        if (grad) {
            gy = at::empty({clips, frames, bins}, target.options());
            check_status(g_api.loss_and_grad(&pr, rows.data_ptr<float>(), (double)pr.B, sot_mean.data_ptr<float>(), nullptr,
                                             (float)(1.0 / (double)pr.B), gy.data_ptr<float>(), nullptr, nullptr, 0, st), p);
            if (w_sot != 1.0) gy.mul_(w_sot);   // the composition's sot_scale_inplace by fl(g * w_sot), g = 1 (the paper's mix: w_sot = 1, nothing to do)
        } else {
            check_status(g_api.loss(&pr, rows.data_ptr<float>(), (double)pr.B, 0, 0.0f, sot_mean.data_ptr<float>(), nullptr, nullptr, nullptr, 0, st), p);
        }

        // MSSLoss * w_mss and its gradient.  The mix weight is applied as the composition applies it -- `loss_fn(a, b) * weight` is ONE float32
        // multiplication of the finished scalar, its backward ONE float32 multiplication of the finished gradient by fl(g * weight) -- inside the
        // finish kernel (post_scale, ABI 13): for the plain `loss.backward()` (g = 1) loss and gradient equal the module-by-module step's bit for bit.
        check_status(g_api.mss(target.data_ptr<float>(), samples, estimate.data_ptr<float>(), samples, clips, samples, sizes, wins, n,
                               (float)mag_weight, (float)logmag_weight, 1e-5f, l2 ? 1 : 0, 0, (float)w_mss, mss_loss.data_ptr<float>(),
                               grad ? grad_audio.data_ptr<float>() : nullptr, mss_ws.data_ptr(), mss_bytes, st), 1.0);
        const at::Tensor& mss_term = mss_loss;
        if (grad) {   // d (w_sot * SOT mean) / d estimate, added to the MSS gradient inside the kernel
            const size_t ws_bytes = g_api.stft_backward_ws(clips, samples, (int)n_fft, (int)hop);
            at::Tensor ws = at::empty({(int64_t)(ws_bytes > 0 ? ws_bytes : 1)}, target.options().dtype(at::kByte));
            check_status(g_api.stft_backward(nullptr, cplx.data_ptr<float>(), clips, samples, samples, window.data_ptr<float>(), (int)n_fft, (int)hop,
                                             gy.data_ptr<float>(), nullptr, grad_audio.data_ptr<float>(), 1, ws.data_ptr(), ws_bytes, st), 1.0);
            ctx->saved_data["grad_audio"] = grad_audio;
        }
        // (total, w_mss * MSSLoss, w_sot * SOT mean): the two terms are what the reference's trainer logs per loss (trainer.py:231-236); values only
        at::Tensor sot_term = w_sot == 1.0 ? sot_mean : at::mul(sot_mean, w_sot);
        at::Tensor total = at::add(mss_term, sot_term);
        ctx->mark_non_differentiable({mss_term, sot_term});
        return {total, mss_term, sot_term};
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grad_outputs)
    {
        at::Tensor g = grad_outputs[0];
        if (g.scalar_type() != at::kFloat) g = g.to(at::kFloat);
        const at::Tensor grad_audio = ctx->saved_data["grad_audio"].toTensor();
        const c10::hip::HIPGuardMasqueradingAsCUDA guard(grad_audio.device());
        at::Tensor out = grad_audio * g;   // the stored gradient is never modified: a retained graph can be walked again
        return {at::Tensor(), out, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(),
                at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

std::vector<at::Tensor> mix_loss_step(const at::Tensor& target, const at::Tensor& estimate, const at::Tensor& window, const at::Tensor& xpos, const at::Tensor& ypos,
                         int64_t n_fft, int64_t hop, double p, int64_t flags, const std::vector<at::Tensor>& mss_windows,
                         const std::vector<int64_t>& mss_sizes, double mag_weight, double logmag_weight, bool l2, double w_mss, double w_sot, bool unit_positions)
{
    TORCH_CHECK(g_api.mss != nullptr && g_api.stft_pair != nullptr, "sot glue: bind() has not been called");
    TORCH_CHECK(!(at::GradMode::is_enabled() && (target.requires_grad() || xpos.requires_grad() || ypos.requires_grad())),
                "sot glue: gradients w.r.t. the target or the positions are not this path's case");
    const bool grad = at::GradMode::is_enabled() && estimate.requires_grad();
    return MixLossStep::apply(target, estimate, window, xpos, ypos, n_fft, hop, p, flags, mss_windows, mss_sizes, mag_weight, logmag_weight, l2, w_mss,
                              w_sot, unit_positions, grad);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "host path of sot_amd.losses.Wasserstein1D in C++ (one call per forward, C++ autograd node); kernels: libsot_hip.so";
    m.def("bind", &bind_library, "dlopen libsot_hip.so at `path` and resolve the entry points; returns its ABI version");
    m.def("mean_loss", &mean_loss, "mean over the rows of W_p^p(x_r, y_r) on planned shared positions; differentiable w.r.t. y");
    m.def("mean_loss_fresh", &mean_loss_fresh, "mean_loss on raw 1-D position tensors: the plan is prepared inside the call (one allocation, one launch)");
    m.def("make_plan", &make_plan, "sot_prepare_positions into one allocation: (sorted x, sorted y, x permutation, y permutation, identity flags); unit: of x / max(x), y / max(y)",
          pybind11::arg("xpos"), pybind11::arg("ypos"), pybind11::arg("unit") = false);
    m.def("audio_to_loss", &audio_to_loss, "STFT magnitudes of target and estimate -> mean SOT loss; differentiable w.r.t. the estimate's audio");
    m.def("stft_magnitude", &stft_magnitude, "[clips, samples] -> [clips, frames, n_fft / 2 + 1] magnitudes (features.TorchSTFT); differentiable w.r.t. the audio");
    m.def("mix_loss_step", &mix_loss_step, "MixOfLosses([MSSLoss, Wasserstein1D]) of the paper's training step on an audio pair: one call, one node -> (total, w_mss * MSSLoss, SOT mean); the total is differentiable w.r.t. the estimate's audio");
    m.def("mss_loss", &mss_loss, "MSSLoss in two launches (sot_mss_loss_and_grad); differentiable w.r.t. the estimate's audio");
}
