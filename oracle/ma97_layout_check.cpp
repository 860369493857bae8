// ma97_layout_check.cpp -- BUILD-TIME check of route B2's struct layouts (TEST / BUILD INFRASTRUCTURE; compiled by oracle/ref_build.mk, never linked
// into the product): the control / info structs of include/mi355x_ma97.h must equal struct ma97_control_d / ma97_info_d of the reference's
// hsl_ma97d.h:68-121 member by member, because a stock Ipopt dlsym()s our ma97_*_d symbols (IpMa97SolverInterface.cpp:303-315) and hands them ITS
// structs.  A drift of either header fails the reference build with the member named.
#include <cstddef>
// the reference header and ours declare the same seven function names with different struct tags: ours are renamed for this translation unit
#define ma97_default_control_d mi355x_decl_default_control
#define ma97_analyse_d         mi355x_decl_analyse
#define ma97_factor_d          mi355x_decl_factor
#define ma97_factor_solve_d    mi355x_decl_factor_solve
#define ma97_solve_d           mi355x_decl_solve
#define ma97_finalise_d        mi355x_decl_finalise
#define ma97_free_akeep_d      mi355x_decl_free_akeep
#include "mi355x_ma97.h"
#undef ma97_default_control_d
#undef ma97_analyse_d
#undef ma97_factor_d
#undef ma97_factor_solve_d
#undef ma97_solve_d
#undef ma97_finalise_d
#undef ma97_free_akeep_d
#include "hsl_ma97d.h"

#define SAME(REF, OURS, m, om) static_assert(offsetof(REF, m) == offsetof(OURS, om) && sizeof(((REF*)0)->m) == sizeof(((OURS*)0)->om), "MA97 layout drift: " #REF "::" #m)
#define C(m) SAME(ma97_control_d, mi355x_ma97_control, m, m)
#define I(m) SAME(ma97_info_d, mi355x_ma97_info, m, m)
static_assert(sizeof(ma97_control_d) == sizeof(mi355x_ma97_control), "MA97 layout drift: sizeof(ma97_control_d)");
static_assert(sizeof(ma97_info_d) == sizeof(mi355x_ma97_info), "MA97 layout drift: sizeof(ma97_info_d)");
C(f_arrays); C(action); C(nemin); C(multiplier); C(ordering); C(print_level); C(scaling);
SAME(ma97_control_d, mi355x_ma97_control, small, small_);
C(u); C(unit_diagnostics); C(unit_error); C(unit_warning); C(factor_min); C(solve_blas3); C(solve_min); C(solve_mf); C(consist_tol); C(ispare); C(rspare);
I(flag); I(flag68); I(flag77); I(matrix_dup); I(matrix_rank); I(matrix_outrange); I(matrix_missing_diag); I(maxdepth); I(maxfront); I(num_delay);
I(num_factor); I(num_flops); I(num_neg); I(num_sup); I(num_two); I(ordering); I(stat); I(maxsupernode); I(ispare); I(rspare);
int mi355x_ma97_layout_checked = 1;
