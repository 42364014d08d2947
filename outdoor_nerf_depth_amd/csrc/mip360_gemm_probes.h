// Component-removal variants of the ping-pong GEMM (mip360_gemm.hip includes this ONLY under -DNERFPP_PROBES; garbage
// results by design, timing experiments of DESIGN.md section 9): -DMIP360_EXP_NODMA / -DMIP360_EXP_NOLDS / -DMIP360_EXP_NOMFMA
#pragma once
namespace mip360 { namespace probe {
#ifdef MIP360_EXP_NODMA
constexpr bool NODMA = true;
#else
constexpr bool NODMA = false;
#endif
#ifdef MIP360_EXP_NOLDS
constexpr bool NOLDS = true;
#else
constexpr bool NOLDS = false;
#endif
#ifdef MIP360_EXP_NOMFMA
constexpr bool NOMFMA = true;
#else
constexpr bool NOMFMA = false;
#endif
}}  // namespace mip360::probe
