// fftconv_binding.cpp -- the reference-side native binding a maintainer would drop in place of csrc/fftconv/fftconv.cpp:
// a torch extension module named `fftconv` with the reference's exact entry points
//     fftconv_fwd(u, filter, D, v, head_dim, q, dropout_mask, gelu, gelu_inp, gelu_q, fft_size, force_fp16_output,
//                 output_hbl_layout, fftfp16)                                  csrc/fftconv/fftconv.cpp:53-61
//     fftconv_bwd(dout, u, filter, D, v, head_dim, q, dropout_mask, gelu, gelu_inp, gelu_q, fft_size,
//                 output_hbl_layout, fftfp16) -> (du, dfilter, dD, dv, dq)     csrc/fftconv/fftconv.cpp:134-142
// implemented on the C ABI of libhyena_fftconv.so (include/hyena_fftconv.h): plain pointers, sizes and a hipStream_t.
// src/ops/fftconv.py of the reference then runs unmodified (it imports `from fftconv import fftconv_fwd, fftconv_bwd`).
// The same translation in Python (ctypes) is overlay/fftconv.py, which is what the CPU tests exercise.
//
// Build (ROCm PyTorch), e.g.:
//   python - <<'PY'
//   from torch.utils.cpp_extension import load
//   load(name="fftconv", sources=["integration/fftconv_binding.cpp"], extra_include_paths=["include", "/opt/rocm/include"],
//        extra_cflags=["-D__HIP_PLATFORM_AMD__"], extra_ldflags=["-Lhyena_dna_amd/csrc", "-lhyena_fftconv",
//        "-Wl,-rpath,$ORIGIN"], with_cuda=False, is_python_module=True)
//   PY
#include <torch/extension.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/core/DeviceGuard.h>

#include <map>
#include <mutex>
#include <tuple>

#include "hyena_fftconv.h"

namespace {

int dtype_code(const torch::Tensor& t) {
    if (t.scalar_type() == torch::kFloat32) return HYENA_F32;
    if (t.scalar_type() == torch::kBFloat16) return HYENA_BF16;
    if (t.scalar_type() == torch::kFloat16) return HYENA_F16;
    TORCH_CHECK(false, "fftconv: u must be float32, bfloat16 or float16");
}

void check(int status) { TORCH_CHECK(status == HYENA_OK, "hyena_fftconv: ", hyena_fftconv_error_string(status)); }

// twiddle tables per (device, transform size, plan), built once (hyena_fftconv_init_tables synchronises: first use only)
torch::Tensor tables_for(const torch::Tensor& like, int L, void* stream) {
    static std::mutex mu;
    static std::map<std::tuple<int, int, int>, torch::Tensor> cache;
    const auto key = std::make_tuple((int)like.get_device(), hyena_fftconv_fft_size(L), hyena_fftconv_plan(L));
    std::lock_guard<std::mutex> g(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    TORCH_CHECK(std::get<1>(key) != 0, "fftconv: unsupported sequence length ", L);
    auto t = torch::empty({(int64_t)hyena_fftconv_table_bytes(L)}, like.options().dtype(torch::kUInt8));
    check(hyena_fftconv_init_tables(t.data_ptr(), L, stream));
    cache[key] = t;
    return t;
}

void refuse_options(const c10::optional<torch::Tensor>& v, int head_dim, const c10::optional<torch::Tensor>& q,
                    const c10::optional<torch::Tensor>& dropout_mask, bool gelu, bool gelu_inp, bool gelu_q, bool output_hbl_layout,
                    bool fftfp16) {
    TORCH_CHECK(!v.has_value() && !q.has_value() && head_dim == 1 && !dropout_mask.has_value() && !gelu && !gelu_inp && !gelu_q &&
                    !output_hbl_layout && !fftfp16,
                "fftconv (MI355X): v / q / head_dim != 1 / dropout_mask / gelu / output_hbl_layout / fftfp16 are not used by any "
                "HyenaDNA configuration and are not implemented");
}

torch::Tensor time_domain_filter(const torch::Tensor& filter, int fft_size, int L) {
    // the reference seam hands over rfft(k, n = fft_size); this library takes k itself
    return torch::fft::irfft(filter, fft_size).slice(-1, 0, L).contiguous();
}

}  // namespace

torch::Tensor fftconv_fwd(torch::Tensor u, torch::Tensor filter, torch::Tensor D, c10::optional<torch::Tensor> v, int head_dim,
                          c10::optional<torch::Tensor> q, c10::optional<torch::Tensor> dropout_mask, bool gelu, bool gelu_inp,
                          bool gelu_q, int fft_size, bool force_fp16_output, bool output_hbl_layout, bool fftfp16) {
    TORCH_CHECK(u.is_cuda() && filter.is_cuda() && D.is_cuda(), "fftconv: tensors must live on the ROCm device");
    refuse_options(v, head_dim, q, dropout_mask, gelu, gelu_inp, gelu_q, output_hbl_layout, fftfp16);
    const int B = u.size(0), H = u.size(1), L = u.size(2);
    TORCH_CHECK(filter.dim() == 2 && filter.size(0) == H && filter.size(1) == fft_size / 2 + 1 && L <= fft_size / 2);
    TORCH_CHECK(D.scalar_type() == torch::kFloat32 && D.numel() == H);
    // PyTorch-ROCm tensors carry the device type "cuda" (HIP masquerading as CUDA): the generic guard and the masquerading
    // stream accessor are the ones that accept it
    c10::DeviceGuard guard(u.device());
    void* stream = (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(u.get_device()).stream();
    u = u.contiguous();
    auto k = time_domain_filter(filter, fft_size, L);
    auto out = torch::empty_like(u);
    auto tab = tables_for(u, L, stream);
    auto ws = torch::empty({(int64_t)hyena_fftconv_workspace_bytes(B, H, L, 0, 0)}, u.options().dtype(torch::kUInt8));
    check(hyena_fftconv_fwd(u.data_ptr(), k.data_ptr<float>(), D.contiguous().data_ptr<float>(), out.data_ptr(), B, H, L, dtype_code(u),
                            tab.data_ptr(), ws.data_ptr(), (size_t)ws.numel(), 0, stream));
    if (force_fp16_output && u.scalar_type() == torch::kFloat32) out = out.to(torch::kFloat16);
    return out;
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, c10::optional<torch::Tensor>, c10::optional<torch::Tensor>>
fftconv_bwd(torch::Tensor dout, torch::Tensor u, torch::Tensor filter, torch::Tensor D, c10::optional<torch::Tensor> v, int head_dim,
            c10::optional<torch::Tensor> q, c10::optional<torch::Tensor> dropout_mask, bool gelu, bool gelu_inp, bool gelu_q,
            int fft_size, bool output_hbl_layout, bool fftfp16) {
    TORCH_CHECK(dout.is_cuda() && u.is_cuda() && filter.is_cuda() && D.is_cuda(), "fftconv: tensors must live on the ROCm device");
    refuse_options(v, head_dim, q, dropout_mask, gelu, gelu_inp, gelu_q, output_hbl_layout, fftfp16);
    const int B = u.size(0), H = u.size(1), L = u.size(2);
    // PyTorch-ROCm tensors carry the device type "cuda" (HIP masquerading as CUDA): the generic guard and the masquerading
    // stream accessor are the ones that accept it
    c10::DeviceGuard guard(u.device());
    void* stream = (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(u.get_device()).stream();
    u = u.contiguous();
    dout = dout.to(u.scalar_type()).contiguous();
    auto k = time_domain_filter(filter, fft_size, L);
    auto du = torch::empty_like(u);
    auto dk = torch::empty({H, L}, u.options().dtype(torch::kFloat32));
    auto dD = torch::empty({H}, u.options().dtype(torch::kFloat32));
    auto tab = tables_for(u, L, stream);
    auto ws = torch::empty({(int64_t)hyena_fftconv_workspace_bytes(B, H, L, 1, 0)}, u.options().dtype(torch::kUInt8));
    check(hyena_fftconv_bwd(dout.data_ptr(), u.data_ptr(), k.data_ptr<float>(), D.contiguous().data_ptr<float>(), du.data_ptr(),
                            dk.data_ptr<float>(), dD.data_ptr<float>(), B, H, L, dtype_code(u), tab.data_ptr(), ws.data_ptr(),
                            (size_t)ws.numel(), 0, stream));
    // src/ops/fftconv.py:98 recovers dk as irfft(dfilter, n = fft_size, norm = 'forward')[..., :L]
    auto dfilter = torch::fft::rfft(dk, fft_size, -1, "forward");
    return std::make_tuple(du, dfilter, dD, c10::optional<torch::Tensor>(), c10::optional<torch::Tensor>());
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("fftconv_fwd", &fftconv_fwd, "Convolution with FFT (MI355X, libhyena_fftconv.so)");
    m.def("fftconv_bwd", &fftconv_bwd, "Convolution with FFT, backward (MI355X, libhyena_fftconv.so)");
}
