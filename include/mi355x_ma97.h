/*
 * mi355x_ma97.h -- route B2 of the drop-in boundary: an HSL_MA97-compatible C symbol
 * set exported by libmi355x_kkt.so, so that a STOCK Ipopt build (no recompilation) loads the MI355X
 * backend with
 *        linear_solver ma97
 *        hsllib        /path/to/libmi355x_kkt.so
 * Ipopt dlopen()s the library (reference src/Common/IpLibraryLoader.cpp:45-75) and dlsym()s exactly
 * these seven symbols (reference src/Algorithm/LinearSolvers/IpMa97SolverInterface.cpp:308-314) with
 * the signatures of IpMa97SolverInterface.hpp:28-99.  The two structs below must match
 * struct ma97_control_d / ma97_info_d of the reference's hsl_ma97d.h:68-121 byte for byte (the
 * field list is an interface fact; the declarations are re-stated here, not copied code).
 */
#ifndef MI355X_MA97_H
#define MI355X_MA97_H
#ifdef __cplusplus
extern "C" {
#endif

struct mi355x_ma97_control {
    int f_arrays; int action; int nemin; double multiplier; int ordering; int print_level; int scaling;
    double small_; double u; int unit_diagnostics; int unit_error; int unit_warning; long factor_min;
    int solve_blas3; long solve_min; int solve_mf; double consist_tol;
    int ispare[5]; double rspare[10];
};
struct mi355x_ma97_info {
    int flag; int flag68; int flag77; int matrix_dup; int matrix_rank; int matrix_outrange; int matrix_missing_diag;
    int maxdepth; int maxfront; int num_delay; long num_factor; long num_flops; int num_neg; int num_sup; int num_two;
    int ordering; int stat; int maxsupernode;
    int ispare[4]; double rspare[10];
};

void ma97_default_control_d(struct mi355x_ma97_control* control);
void ma97_analyse_d(int check, int n, const int ptr[], const int row[], double val[], void** akeep,
                    const struct mi355x_ma97_control* control, struct mi355x_ma97_info* info, int order[]);
void ma97_factor_d(int matrix_type, const int ptr[], const int row[], const double val[], void** akeep, void** fkeep,
                   const struct mi355x_ma97_control* control, struct mi355x_ma97_info* info, double scale[]);
void ma97_factor_solve_d(int matrix_type, const int ptr[], const int row[], const double val[], int nrhs, double x[], int ldx,
                         void** akeep, void** fkeep, const struct mi355x_ma97_control* control, struct mi355x_ma97_info* info,
                         double scale[]);
void ma97_solve_d(int job, int nrhs, double* x, int ldx, void** akeep, void** fkeep,
                  const struct mi355x_ma97_control* control, struct mi355x_ma97_info* info);
void ma97_finalise_d(void** akeep, void** fkeep);
void ma97_free_akeep_d(void** akeep);

#ifdef __cplusplus
}
#endif
#endif
