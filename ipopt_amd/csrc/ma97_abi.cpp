// ma97_abi.cpp -- HSL_MA97-compatible exports on top of the native C ABI (include/mi355x_ma97.h).
// Call protocol honoured = what the reference's MA97 adapter does (IpMa97SolverInterface.cpp):
//   default_control -> [adapter sets f_arrays=1, action=0, nemin, small, u, ordering, scaling] (:326-338)
//   analyse(check=0, n, ptr=ia, row=ja, val=NULL|vals, &akeep, &control, &info, order=NULL) (:567, :674)
//   factor(matrix_type=4, ptr,row,val,&akeep,&fkeep,&control,&info,scale) (:707) -> info.flag (7/-7 singular :719,
//   <0 fatal :773), info.num_neg (:779), info.num_delay, info.matrix_rank
//   solve(job=0,nrhs,x,ldx=n,...) (:790,:805) ; finalise (:301)
// Input is 1-based (f_arrays) CSC-lower == CSR-upper of the KKT matrix (IpMa97SolverInterface.hpp:249).
#include "../../include/mi355x_ma97.h"
#include "../../include/mi355x_kkt.h"
#include <cfloat>
#include <cstring>
#include <new>
#include <vector>

namespace {
struct Keep {
    mi355x_kkt_handle h = nullptr;
    int n = 0, nnz = 0, base = 1;
    std::vector<int> ptr, row;
    bool analysed = false;
    bool with_values = false;   // the analysis in place saw values (zero-diagonal 2x2 pre-pairing done)
    double u = -1.0;
};

void fill_info(Keep* k, mi355x_ma97_info* info)
{
    mi355x_kkt_info I;
    if (!k->h || mi355x_kkt_get_info(k->h, &I) != 0) return;
    info->maxdepth = I.num_levels; info->maxfront = I.maxfront; info->maxsupernode = I.maxsupernode;
    info->num_factor = (long)I.nnz_l; info->num_flops = (long)I.flops_factor; info->num_sup = I.num_sn;
    info->num_neg = I.num_neg; info->num_two = I.num_two; info->num_delay = I.num_delayed + I.num_small;      /* columns moved to a parent front + what had to be forced */ info->matrix_rank = I.n - I.num_zero;
}

bool do_analyse(Keep* k, const mi355x_ma97_control* c, const double* val, mi355x_ma97_info* info)
{
    if (!k->h) {
        mi355x_kkt_options o; mi355x_kkt_default_options(&o);
        o.index_base = k->base;
        if (c) { if (c->nemin > 0) o.nemin = c->nemin; if (c->u > 0) o.pivtol = c->u; if (c->small_ > 0) o.small = c->small_;
                 o.ordering = (c->ordering == 1 || c->ordering == 2) ? 1 : 0; o.verbose = c->print_level > 0 ? 1 : 0; }
        if (mi355x_kkt_create(&k->h, &o) != 0) { info->flag = -1; info->stat = 1; return false; }
    }
    if (mi355x_kkt_analyse(k->h, k->n, k->nnz, k->ptr.data(), k->row.data(), MI355X_KKT_FMT_CSR_UPPER, val) != 0) { info->flag = -4; return false; }
    k->analysed = true; k->with_values = (val != nullptr);
    info->ordering = (c && (c->ordering == 1 || c->ordering == 2)) ? 1 : 3;
    fill_info(k, info);
    return true;
}
}  // namespace

extern "C" {

void ma97_default_control_d(mi355x_ma97_control* c)
{
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->f_arrays = 0; c->action = 1; c->nemin = 8; c->multiplier = 1.1; c->ordering = 5; c->print_level = 0; c->scaling = 0;
    c->small_ = 1e-20; c->u = 0.01; c->unit_diagnostics = 6; c->unit_error = 6; c->unit_warning = 6;
    c->factor_min = 20000000L; c->solve_blas3 = 0; c->solve_min = 100000L; c->solve_mf = 0; c->consist_tol = DBL_EPSILON;
}

void ma97_analyse_d(int /*check*/, int n, const int ptr[], const int row[], double val[], void** akeep,
                    const mi355x_ma97_control* control, mi355x_ma97_info* info, int order[])
{
    if (!info) return;
    std::memset(info, 0, sizeof(*info));
    if (!akeep || !ptr || !row || n < 0) { info->flag = -2; return; }
    try {
        Keep* k = static_cast<Keep*>(*akeep);
        if (!k) { k = new Keep(); *akeep = k; }
        k->base = (control && control->f_arrays) ? 1 : 0;
        k->n = n; k->nnz = ptr[n] - k->base;
        k->ptr.assign(ptr, ptr + n + 1); k->row.assign(row, row + k->nnz);
        k->analysed = false;
        info->matrix_rank = n;
        // The analysis runs NOW, also without values: errors (index out of range, out of memory) surface here and
        // info.num_factor / num_flops / maxfront are the real predictions -- the adapter's `ma97_order best` compares
        // info.num_flops of two analyses (IpMa97SolverInterface.cpp:567-600).  Without values the zero-diagonal 2x2
        // pre-pairing cannot be done, so the first factor call redoes the analysis once with them (akeep is opaque to the
        // caller).  With values (the matching-based orderings 7/8, IpMa97SolverInterface.cpp:654-674) this one is final.
        if (!do_analyse(k, control, val, info)) return;
        if (order) { for (int i = 0; i < n; ++i) order[i] = i + k->base; }
        if (order && k->analysed) {
            std::vector<int> perm(n);
            if (mi355x_kkt_get_symbolic(k->h, 0, perm.data(), n) == 0)
                for (int newi = 0; newi < n; ++newi) order[perm[newi]] = newi + k->base;   // order[i] = position of variable i
        }
    } catch (const std::bad_alloc&) { info->flag = -1; info->stat = 1; } catch (...) { info->flag = -1; }
}

void ma97_factor_d(int /*matrix_type*/, const int /*ptr*/[], const int /*row*/[], const double val[], void** akeep, void** fkeep,
                   const mi355x_ma97_control* control, mi355x_ma97_info* info, double scale[])
{
    if (!info) return;
    std::memset(info, 0, sizeof(*info));
    if (!akeep || !*akeep || !val) { info->flag = -2; return; }
    try {
        Keep* k = static_cast<Keep*>(*akeep);
        if ((!k->analysed || !k->with_values) && !do_analyse(k, control, val, info)) return;
        if (control && control->u > 0 && control->u != k->u) { mi355x_kkt_set_pivtol(k->h, control->u > 0.5 ? 0.5 : control->u); k->u = control->u; }
        // scaling semantics of the MA97 call protocol (IpMa97SolverInterface.cpp:641-652,707):
        //   control.scaling  > 0               compute factors (matching scaling for MC64, Ruiz equilibration on the device for
        //                                      MC77 / MC30) and WRITE them to scale[n]
        //   control.scaling == 0, scale given  apply the caller-held factors ("reuse")
        //   control.scaling == 0, scale NULL   no scaling
        // HSL's choices map onto ours: 1 (MC64) and 3 (MC64 from the matching ordering) -> matching scaling; 2 (MC77) and 4 (MC30) ->
        // Ruiz equilibration (MC77 IS Ruiz's algorithm)
        // (MI355X_KKT_MA97_MATCHING=device: the matching scaling is computed by the device auction, scaling mode 5, instead of the host algorithm)
        static const int match_mode = [] { const char* e = getenv("MI355X_KKT_MA97_MATCHING"); return (e && std::strcmp(e, "device") == 0) ? 5 : 3; }();
        const int mode = (control && control->scaling > 0) ? ((control->scaling == 1 || control->scaling == 3) ? match_mode : 1) : (scale ? 2 : 0);
        if (mi355x_kkt_set_scaling(k->h, mode, scale) != 0) { info->flag = -1; return; }
        double* buf = mi355x_kkt_values_buffer(k->h);
        if (!buf) { info->flag = -1; return; }
        std::memcpy(buf, val, sizeof(double) * (size_t)k->nnz);
        int nneg = 0, nzero = 0;
        int st = mi355x_kkt_factor(k->h, nullptr, &nneg, &nzero);
        if (fkeep) *fkeep = k;
        fill_info(k, info);
        if (st == MI355X_KKT_FATAL) { info->flag = -1; return; }
        if (scale && (mode == 1 || mode == 3 || mode == 5) && mi355x_kkt_get_scaling(k->h, scale) != 0) { info->flag = -1; return; }   // hand the factors back for reuse
        if (st == MI355X_KKT_SINGULAR) info->flag = (control && control->action) ? 7 : -7;
        else info->flag = 0;
    } catch (...) { info->flag = -1; }
}

void ma97_solve_d(int job, int nrhs, double* x, int ldx, void** akeep, void** /*fkeep*/,
                  const mi355x_ma97_control* /*control*/, mi355x_ma97_info* info)
{
    if (!info) return;
    info->flag = 0;
    if (!akeep || !*akeep || !x) { info->flag = -2; return; }
    if (job != 0) { info->flag = -12; return; }   // partial solves are not offered through this route
    Keep* k = static_cast<Keep*>(*akeep);
    if (mi355x_kkt_solve(k->h, nrhs, x, ldx) != 0) info->flag = -1;
}

void ma97_factor_solve_d(int matrix_type, const int ptr[], const int row[], const double val[], int nrhs, double x[], int ldx,
                         void** akeep, void** fkeep, const mi355x_ma97_control* control, mi355x_ma97_info* info, double scale[])
{
    ma97_factor_d(matrix_type, ptr, row, val, akeep, fkeep, control, info, scale);
    if (!info || info->flag < 0) return;
    int flag = info->flag;
    ma97_solve_d(0, nrhs, x, ldx, akeep, fkeep, control, info);
    if (info->flag == 0) info->flag = flag;
}

void ma97_free_akeep_d(void** akeep)
{
    if (!akeep || !*akeep) return;
    Keep* k = static_cast<Keep*>(*akeep);
    try { if (k->h) mi355x_kkt_destroy(k->h); delete k; } catch (...) {}
    *akeep = nullptr;
}

void ma97_finalise_d(void** akeep, void** fkeep)
{
    if (fkeep) *fkeep = nullptr;   // fkeep aliases akeep's object
    ma97_free_akeep_d(akeep);
}

}  // extern "C"
