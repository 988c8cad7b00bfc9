// Stand-alone check of rocFFT 2-D real forward transforms, no code of this repository:
// transform A = (Fy, Fx) then B, each against a direct DFT on the host.  Used to decide
// whether an order dependence between (30, 120) and (60, 60) is rocFFT's or ours.
//   hipcc --offload-arch=gfx950 repro.cpp -lrocfft -o repro && ./repro 30 120 60 60
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

static double run(int Fy, int Fx, int count) {
    const int Fxh = Fx / 2 + 1;
    std::vector<float> in((size_t)count * Fy * Fx);
    srand(Fy * 1000 + Fx);
    for (auto &v : in) v = (float)rand() / RAND_MAX - 0.5f;
    float *d_in;
    float2 *d_out;
    hipMalloc((void **)&d_in, in.size() * sizeof(float));
    hipMalloc((void **)&d_out, (size_t)count * Fy * Fxh * sizeof(float2));
    hipMemcpy(d_in, in.data(), in.size() * sizeof(float), hipMemcpyHostToDevice);
    rocfft_plan plan = nullptr;
    const size_t lengths[2] = {(size_t)Fx, (size_t)Fy};
    rocfft_plan_create(&plan, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                       rocfft_precision_single, 2, lengths, (size_t)count, nullptr);
    size_t wb = 0;
    rocfft_plan_get_work_buffer_size(plan, &wb);
    rocfft_execution_info info = nullptr;
    rocfft_execution_info_create(&info);
    void *work = nullptr;
    if (wb) {
        hipMalloc(&work, wb);
        rocfft_execution_info_set_work_buffer(info, work, wb);
    }
    void *ins[1] = {d_in}, *outs[1] = {d_out};
    rocfft_execute(plan, ins, outs, info);
    hipDeviceSynchronize();
    std::vector<float2> out((size_t)count * Fy * Fxh);
    hipMemcpy(out.data(), d_out, out.size() * sizeof(float2), hipMemcpyDeviceToHost);
    double worst = 0, scale = 0;
    const int img = count - 1;
    for (int ky = 0; ky < Fy; ky += 7)
        for (int kx = 0; kx < Fxh; kx += 5) {
            std::complex<double> s = 0;
            for (int y = 0; y < Fy; ++y)
                for (int x = 0; x < Fx; ++x) {
                    const double ph = -2 * M_PI * ((double)ky * y / Fy + (double)kx * x / Fx);
                    s += (double)in[((size_t)img * Fy + y) * Fx + x] * std::complex<double>(cos(ph), sin(ph));
                }
            const float2 g = out[((size_t)img * Fy + ky) * Fxh + kx];
            worst = fmax(worst, std::abs(s - std::complex<double>(g.x, g.y)));
            scale = fmax(scale, std::abs(s));
        }
    // inverse: back to real, must reproduce the input times Fy * Fx
    rocfft_plan inv = nullptr;
    rocfft_plan_create(&inv, rocfft_placement_notinplace, rocfft_transform_type_real_inverse,
                       rocfft_precision_single, 2, lengths, (size_t)count, nullptr);
    size_t wb2 = 0;
    rocfft_plan_get_work_buffer_size(inv, &wb2);
    rocfft_execution_info info2 = nullptr;
    rocfft_execution_info_create(&info2);
    void *work2 = nullptr;
    if (wb2) {
        hipMalloc(&work2, wb2);
        rocfft_execution_info_set_work_buffer(info2, work2, wb2);
    }
    float *d_back;
    hipMalloc((void **)&d_back, in.size() * sizeof(float));
    void *ins2[1] = {d_out}, *outs2[1] = {d_back};
    rocfft_execute(inv, ins2, outs2, info2);
    hipDeviceSynchronize();
    std::vector<float> back(in.size());
    hipMemcpy(back.data(), d_back, back.size() * sizeof(float), hipMemcpyDeviceToHost);
    double rt = 0;
    for (size_t i = 0; i < in.size(); ++i) rt = fmax(rt, fabs(back[i] / ((double)Fy * Fx) - in[i]));
    printf("   round trip error %.2e\n", rt);
    hipFree(d_back);
    if (work2) hipFree(work2);
    rocfft_execution_info_destroy(info2);
    rocfft_execution_info_destroy(info);
    if (!getenv("KEEP_PLANS")) {
        rocfft_plan_destroy(plan);
        rocfft_plan_destroy(inv);
    }
    if (work) hipFree(work);
    hipFree(d_in);
    hipFree(d_out);
    return worst / scale;
}

int main(int argc, char **argv) {
    rocfft_setup();
    for (int i = 1; i + 1 < argc; i += 2) {
        const int Fy = atoi(argv[i]), Fx = atoi(argv[i + 1]);
        printf("real forward %d x %d, 2 transforms: relative error %.2e\n", Fy, Fx, run(Fy, Fx, 2));
    }
    rocfft_cleanup();
    return 0;
}
