// env_knobs.h -- the two development variables behind which every switch and tunable of the library sits (round 6: ~45 separate
// MI355X_KKT_NO_* / MI355X_KKT_*_MIN / ... variables folded into two; the measured-and-rejected code paths they selected are deleted).
//
//   MI355X_KKT_DISABLE=name[,name...]   switches a KEPT fast path off, so that the tests can compare it bitwise with the plain path underneath:
//       lookahead chain_solve fuse_dt fastpiv asm_pull pair_solve selfasm xcd_tiles xcd_affine fuse_upd grouped tfuse leafchain side_small
//       front_df p1_small norestore optimistic subcomm blockcache thread_pool purify solve_ctx keep_scale
//   MI355X_KKT_TUNE=name=value[,...]    numeric thresholds of the schedule (defaults are the measured optima; tests force a path with them):
//       la_wgs la_min_nt la_min_tiles grp_rbw_max grp_maxchains fuse_dt_maxwg chain_solve_maxc fastpiv_floor
// Both are read when a handle is set up (analyse / restructure), `optimistic` at the first factorisation of the process.
#pragma once
#include <cstdlib>
#include <cstring>
#include <string>

namespace mi355x {

inline std::string knob_env(const char* var)      // (read at every query -- set-up time only: a test switches a path off for ONE handle of the process)
{
    const char* e = getenv(var[0] == 'D' ? "MI355X_KKT_DISABLE" : "MI355X_KKT_TUNE");
    return std::string(",") + (e ? e : "") + ",";
}
inline bool knob_disabled(const char* name)
{
    const std::string s = knob_env("D");
    return s.find(std::string(",") + name + ",") != std::string::npos;
}
inline bool knob_tune(const char* name, double* value)
{
    const std::string s = knob_env("T");
    const std::string key = std::string(",") + name + "=";
    const size_t p = s.find(key);
    if (p == std::string::npos) return false;
    *value = atof(s.c_str() + p + key.size());
    return true;
}
// MI355X_KKT_TRACE=item[,item...]   profiling hooks: clocks (phase stamps of the pivot-block kernels, tools/clocks.py), launches (name the launch whose configuration the
// runtime refuses), matching (phase times of the host matching scaling), solve=<file> (time line of the data-flow solve sweeps, tools/solve_trace.py)
inline bool knob_trace(const char* name, std::string* value = nullptr)
{
    const char* e = getenv("MI355X_KKT_TRACE");
    const std::string s = std::string(",") + (e ? e : "") + ",";
    size_t p = s.find(std::string(",") + name + ",");
    if (p != std::string::npos) { if (value) value->clear(); return true; }
    const std::string key = std::string(",") + name + "=";
    p = s.find(key);
    if (p == std::string::npos) return false;
    if (value) *value = s.substr(p + key.size(), s.find(',', p + key.size()) - p - key.size());
    return true;
}
inline long long knob_int(const char* name, long long dflt) { double v; return knob_tune(name, &v) ? (long long)v : dflt; }

}  // namespace mi355x
