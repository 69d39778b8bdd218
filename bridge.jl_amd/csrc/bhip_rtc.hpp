// bhip_rtc.hpp -- user-defined target drift b(t,x,P) compiled at run time with hipRTC.
//
// The reference lets users define a process by adding Julia methods Bridge.b / Bridge.sigma on their own
// type (README.md:69-77, project_partialbridge/partialbridge_fitzhugh.jl:44-46).  A Julia closure
// cannot run inside a pre-compiled kernel, so besides the registry of built-in functors a
// user can hand the library the BODY of the drift as HIP C++ text:
//
//     "o[0] = (x[0] - x[1] - x[0]*x[0]*x[0] + par[1]) / par[0];  o[1] = par[2]*x[0] - x[1] + par[3];"
//
// (inputs: double t, const double* x, const double* par; output double* o).  The diffusion coefficient
// is constant on this path and is passed as DATA (a d x m' matrix appended to the
// parameters).  The text is spliced into a functor next to the embedded kernel source and every
// kernel instantiation that is actually launched is compiled for gfx950 on first use and cached.
#pragma once
#include "bhip_path_kernel.h"
#include <hip/hiprtc.h>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace bhip {

static const char *const RTC_PREFIX =
#include "bhip_rtc_src.inc"
    ;

static const char *const RTC_PREFIX_TILE =
#include "bhip_rtc_tile_src.inc"
    ;

struct UserModel {
    int id = 0, d = 0, mp = 0, npar = 0;
    bool components = false;   // d > 3: the drift is given component-wise and runs on the MFMA tile kernel (bhip_model_define_components)
    std::string drift;
    std::string sigma;   // empty: constant sigma passed as data; else the body of sigma(t,x,P)
    std::map<std::vector<int>, hipFunction_t> fns;   // (device, gk, mo, noise, fl) -> kernel
    std::vector<hipModule_t> modules;
};

inline std::vector<std::unique_ptr<UserModel>> &user_models()
{
    static std::vector<std::unique_ptr<UserModel>> v;
    return v;
}
inline std::mutex &user_models_mutex()
{
    static std::mutex m;
    return m;
}
constexpr int USER_MODEL_BASE = 1000;

inline UserModel *find_user_model(int id)
{
    const int k = id - USER_MODEL_BASE;
    auto &v = user_models();
    return (k >= 0 && k < (int)v.size()) ? v[k].get() : nullptr;
}

// the instantiation behind (noise, npair): npair > 0 selects the wave-specialised kernel k_pc (noise 6 / 7 = NOISE_FRESH_PC /
// NOISE_PCN_LINES_PC of bhip_pc_kernel.h) with that many producer/consumer pairs per workgroup (npair < 0: see below)
inline std::string rtc_kernel_name(int gk, int mo, int noise, int fl, int npair, bool qualified)
{
    const std::string ns = qualified ? "bhip::" : "";
    const std::string args = std::to_string(gk) + ", " + std::to_string(mo) + ", ";
    if (npair > 0) return ns + "k_pc<" + ns + "MUser, " + args + std::to_string(noise) + ", " + std::to_string(fl) + ", " + std::to_string(npair) + ">";
    if (npair < 0)   // -n: n pairs WITHOUT the coefficient rows in LDS (the large-ensemble workgroup of the noise specification v4)
        return ns + "k_pc<" + ns + "MUser, " + args + std::to_string(noise) + ", " + std::to_string(fl) + ", " + std::to_string(-npair) + ", false, false>";
    if (noise == NOISE_PCN_LINES) return ns + "k_chain_lines<" + ns + "MUser, " + args + std::to_string(fl) + ">";
    return ns + "k_paths<" + ns + "MUser, " + args + std::to_string(noise) + ", " + std::to_string(fl) + ">";
}

// component-wise drift at 4 <= d <= 8: one path per lane, MUser STREAMED like MLinPro<D, bhip_cptr_t> (bhip_models.h) -- the device block
// [drift parameters | sigma | a | inv(sigma)] is read through the scalar unit, every matrix in column blocks behind the phase before it
inline std::string rtc_source_components(const UserModel &um, int gk, int mo, int noise, int fl)
{
    std::string s = RTC_PREFIX;
    const std::string D = std::to_string(um.d), NP = std::to_string(um.npar);
    s += "\nnamespace bhip {\nstruct MUser {\n";
    s += "    static constexpr int D = " + D + ", MP = " + D + ", NP = " + NP + ", ID = 1000;\n";
    s += "    static constexpr bool STREAMED = true;\n    static constexpr bool noisy(int) { return true; }\n    bhip_cptr_t p;\n";
    s += "    BHIP_DEV explicit MUser(bhip_cptr_t p_) : p(p_) {}\n";
    s += "    static BHIP_DEV double bk(int k, double t, const double *x, const double *par)\n    {\n";
    s += "        const int d = D; (void)d; (void)t; (void)par; (void)k;\n        double o = 0.0;\n        " + um.drift + "\n        return o;\n    }\n";
    s += R"(    BHIP_DEV void b(double t, const double *x, double *o) const
    {
        bhip_cptr_t q = p;
        bhip_after(q, x[D - 1]);
        double par[NP > 0 ? NP : 1];
#pragma unroll
        for (int k = 0; k < NP; k++) par[k] = q[k];
#pragma unroll
        for (int k = 0; k < D; k++) o[k] = bk(k, t, x, par);   // (k is a constant after unrolling: x stays in registers)
    }
    BHIP_DEV void sdw(double, const double *, const double *dw, double *o) const
    {
        bhip_cptr_t S = p + NP;
        bhip_after(S, dw[D - 1]);
        matvec_streamed<D, bhip_cptr_t>(S, dw, o);
    }
    BHIP_DEV void amul(double, const double *, const double *r, double *o) const
    {
        bhip_cptr_t A = p + NP + D * D;
        bhip_after(A, r[D - 1]);
        matvec_streamed<D, bhip_cptr_t>(A, r, o);
    }
    BHIP_DEV void sinv_mul(const double *v, double *o) const
    {
        bhip_cptr_t Sq = p + NP + 2 * D * D;
        bhip_after(Sq, v[D - 1]);
        matvec_streamed<D, bhip_cptr_t>(Sq, v, o);
    }
};
)";
    s += "template __global__ void " + rtc_kernel_name(gk, mo, noise, fl, 0, false) + "(const KArgs);\n}\n";
    return s;
}

inline std::string rtc_source(const UserModel &um, int gk, int mo, int noise, int fl, int npair = 0)
{
    if (um.components) return rtc_source_components(um, gk, mo, noise, fl);
    std::string s = RTC_PREFIX;
    s += "\nnamespace bhip {\nstruct MUser {\n";
    s += "    static constexpr int D = " + std::to_string(um.d) + ", MP = " + std::to_string(um.mp) + ", NP = " + std::to_string(um.npar) + ", ID = 1000;\n";
    s += "    static constexpr bool noisy(int) { return true; }\n    const double *p;\n";
    s += "    BHIP_DEV explicit MUser(const double *p_) : p(p_) {}\n";
    s += "    BHIP_DEV void b(double t, const double *x, double *o) const\n    {\n        const double *par = p; (void)par; (void)t;\n        " + um.drift + "\n    }\n";
    if (um.sigma.empty()) {
        // constant sigma passed as data behind the drift parameters: sigma (D x MP), a = sigma*sigma' (D x D), [inv(sigma)]
        s += R"(    BHIP_DEV void sdw(double, const double *, const double *dw, double *o) const
    {
        const double *S = p + NP;
#pragma unroll
        for (int i = 0; i < D; i++) {
            double s = S[i] * dw[0];
#pragma unroll
            for (int j = 1; j < MP; j++) s += S[i + D * j] * dw[j];
            o[i] = s;
        }
    }
    BHIP_DEV void amul(double, const double *, const double *r, double *o) const
    {
        const double *A = p + NP + D * MP;
#pragma unroll
        for (int i = 0; i < D; i++) {
            double s = A[i] * r[0];
#pragma unroll
            for (int j = 1; j < D; j++) s += A[i + D * j] * r[j];
            o[i] = s;
        }
    }
)";
        if (um.d == um.mp)   // square sigma: inv(sigma)*v for innovations!
            s += R"(    BHIP_DEV void sinv_mul(const double *v, double *o) const
    {
        const double *Si = p + NP + D * MP + D * D;
#pragma unroll
        for (int i = 0; i < D; i++) {
            double s = Si[i] * v[0];
#pragma unroll
            for (int j = 1; j < D; j++) s += Si[i + D * j] * v[j];
            o[i] = s;
        }
    }
)";
    } else {
        // state-dependent sigma(t,x,P): the user text fills s (D x MP, column-major, zero-initialised);
        // a(t,x,P) = sigma*sigma' (src/types.jl:32), constdiff(P) = false
        s += "    static constexpr bool STATE_SIGMA = true;\n";
        s += "    BHIP_DEV void sig(double t, const double *x, double *s) const\n    {\n        const double *par = p; (void)par; (void)t; (void)x;\n";
        s += "#pragma unroll\n        for (int k = 0; k < D * MP; k++) s[k] = 0.0;\n        " + um.sigma + "\n    }\n";
        s += R"(    BHIP_DEV void sdw(double t, const double *x, const double *dw, double *o) const
    {
        double S[D * MP];
        sig(t, x, S);
#pragma unroll
        for (int i = 0; i < D; i++) {
            double s = S[i] * dw[0];
#pragma unroll
            for (int j = 1; j < MP; j++) s += S[i + D * j] * dw[j];
            o[i] = s;
        }
    }
    BHIP_DEV void amat(double t, const double *x, double *A) const
    {
        double S[D * MP];
        sig(t, x, S);
#pragma unroll
        for (int i = 0; i < D; i++)
#pragma unroll
            for (int j = 0; j < D; j++) {
                double s = S[i] * S[j];
#pragma unroll
                for (int k = 1; k < MP; k++) s += S[i + D * k] * S[j + D * k];
                A[i + D * j] = s;
            }
    }
    BHIP_DEV void amul(double t, const double *x, const double *r, double *o) const
    {
        double A[D * D];
        amat(t, x, A);
#pragma unroll
        for (int i = 0; i < D; i++) {
            double s = A[i] * r[0];
#pragma unroll
            for (int j = 1; j < D; j++) s += A[i + D * j] * r[j];
            o[i] = s;
        }
    }
)";
    }
    s += "};\n";
    s += "template __global__ void " + rtc_kernel_name(gk, mo, noise, fl, npair, false) + "(const KArgs);\n}\n";
    return s;
}

// compile one instantiation (no GPU needed); returns "" on success, else the hipRTC log
inline std::string rtc_compile(const UserModel &um, int gk, int mo, int noise, int fl, std::vector<char> &code, std::string &low, int npair = 0)
{
    const std::string src = rtc_source(um, gk, mo, noise, fl, npair);
    const std::string name = rtc_kernel_name(gk, mo, noise, fl, npair, true);
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "bhip_user_model.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return "hiprtcCreateProgram failed";
    hiprtcAddNameExpression(prog, name.c_str());
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off"};
    const hiprtcResult r = hiprtcCompileProgram(prog, 4, opts);
    if (r != HIPRTC_SUCCESS) {
        size_t n = 0;
        hiprtcGetProgramLogSize(prog, &n);
        std::string log(n, ' ');
        if (n) hiprtcGetProgramLog(prog, &log[0]);
        hiprtcDestroyProgram(&prog);
        return "hipRTC compilation of the user process failed:\n" + log;
    }
    const char *lowered = nullptr;
    hiprtcGetLoweredName(prog, name.c_str(), &lowered);
    low = lowered ? lowered : "";
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    code.resize(cs);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    return "";
}

// ---- d > 3: k_tile<D, NOISE, PAD, MUserBig>, the target drift evaluated component-wise by the user's text
inline std::string rtc_tile_compile(const UserModel &um, int D, int noise, bool pad, std::vector<char> &code, std::string &low, bool tda = false)
{
    std::string s = RTC_PREFIX_TILE;
    s += "\nnamespace bhip {\nstruct MUserBig {\n    static constexpr bool ON = true;\n";
    s += "    static __device__ __forceinline__ double bk(int k, double t, const double *x, const double *par)\n    {\n";
    s += "        const int d = " + std::to_string(um.d) + "; (void)d; (void)t; (void)par; (void)k;\n        double o = 0.0;\n        " + um.drift + "\n        return o;\n    }\n};\n";
    const std::string tail = tda ? ", true>" : ", false>";   // TDA: a per-step auxiliary matrix beside the user drift (bhip_tile_kernel.h)
    const std::string inst = "k_tile<" + std::to_string(D) + ", " + std::to_string(noise) + ", " + (pad ? "true" : "false") + ", MUserBig" + tail;
    s += "template __global__ void " + inst + "(const TArgs);\n}\n";
    const std::string name = "bhip::k_tile<" + std::to_string(D) + ", " + std::to_string(noise) + ", " + (pad ? "true" : "false") + ", bhip::MUserBig" + tail;
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, s.c_str(), "bhip_user_model_tile.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return "hiprtcCreateProgram failed";
    hiprtcAddNameExpression(prog, name.c_str());
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off"};
    const hiprtcResult r = hiprtcCompileProgram(prog, 4, opts);
    if (r != HIPRTC_SUCCESS) {
        size_t n = 0;
        hiprtcGetProgramLogSize(prog, &n);
        std::string log(n, ' ');
        if (n) hiprtcGetProgramLog(prog, &log[0]);
        hiprtcDestroyProgram(&prog);
        return "hipRTC compilation of the user process failed:\n" + log;
    }
    const char *lowered = nullptr;
    hiprtcGetLoweredName(prog, name.c_str(), &lowered);
    low = lowered ? lowered : "";
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    code.resize(cs);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    return "";
}
inline std::string rtc_tile_build(UserModel &um, int D, int noise, bool pad, hipFunction_t *out, bool tda = false)
{
    std::vector<char> code;
    std::string low;
    const std::string log = rtc_tile_compile(um, D, noise, pad, code, low, tda);
    if (!log.empty()) return log;
    hipModule_t mod;
    if (hipModuleLoadData(&mod, code.data()) != hipSuccess) return "hipModuleLoadData failed for the compiled user model";
    hipFunction_t f;
    if (hipModuleGetFunction(&f, mod, low.c_str()) != hipSuccess) return "kernel symbol not found in the compiled user model: " + low;
    um.modules.push_back(mod);
    *out = f;
    return "";
}

// compile + load one instantiation
inline std::string rtc_build(UserModel &um, int gk, int mo, int noise, int fl, hipFunction_t *out, int npair = 0)
{
    std::vector<char> code;
    std::string low;
    const std::string log = rtc_compile(um, gk, mo, noise, fl, code, low, npair);
    if (!log.empty()) return log;
    hipModule_t mod;
    if (hipModuleLoadData(&mod, code.data()) != hipSuccess) return "hipModuleLoadData failed for the compiled user model";
    hipFunction_t f;
    if (hipModuleGetFunction(&f, mod, low.c_str()) != hipSuccess) return "kernel symbol not found in the compiled user model: " + low;
    um.modules.push_back(mod);
    *out = f;
    return "";
}

}  // namespace bhip
