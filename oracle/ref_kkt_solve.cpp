// ref_kkt_solve.cpp -- TEST / BASELINE INFRASTRUCTURE: times the REFERENCE's own CPU linear-solver path
// (TripletToCSRConverter + PardisoMKLSolverInterface from the unmodified reference, oneMKL PARDISO underneath;
// call sites IpPardisoMKLSolverInterface.cpp:440-715, converter IpTripletToCSRConverter.cpp:46-372) on a KKT
// system given as a binary triplet file.  Used by bench.py's `cpu_baseline` leg ("kind": "reference").
//
// file format (little endian): int32 n, int32 nnz, int32 irn[nnz], int32 jcn[nnz] (1-based), double a[nnz], double rhs[n]
// usage: ref_kkt_solve <file> <nfactor> <nsolve_per_factor>       -> prints one JSON line
#include "IpIpoptApplication.hpp"
#include "IpPardisoMKLSolverInterface.hpp"
#include "IpTripletToCSRConverter.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

using namespace Ipopt;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
   if( argc < 4 ) { fprintf(stderr, "usage: %s file nfactor nsolve\n", argv[0]); return 2; }
   FILE* f = fopen(argv[1], "rb");
   if( !f ) { perror("open"); return 2; }
   int n, nnz;
   if( fread(&n, 4, 1, f) != 1 || fread(&nnz, 4, 1, f) != 1 ) return 2;
   std::vector<Index> irn(nnz), jcn(nnz);
   std::vector<Number> a(nnz), rhs(n);
   if( fread(irn.data(), 4, nnz, f) != (size_t) nnz || fread(jcn.data(), 4, nnz, f) != (size_t) nnz ||
       fread(a.data(), 8, nnz, f) != (size_t) nnz || fread(rhs.data(), 8, n, f) != (size_t) n ) return 2;
   fclose(f);
   int nfactor = atoi(argv[2]), nsolve = atoi(argv[3]);

   SmartPtr<IpoptApplication> app = IpoptApplicationFactory();
   app->Options()->SetIntegerValue("print_level", 0);
   if( getenv("REF_PARDISO_MSGLVL") ) app->Options()->SetIntegerValue("pardisomkl_msglvl", atoi(getenv("REF_PARDISO_MSGLVL")));   // MKL's own statistics (nnz(L), flops)
   app->Initialize("");
   SmartPtr<PardisoMKLSolverInterface> iface = new PardisoMKLSolverInterface();
   if( !iface->ReducedInitialize(*app->Jnlst(), *app->Options(), "") ) { fprintf(stderr, "ReducedInitialize failed\n"); return 3; }
   // what TSymLinearSolver does for a CSR_Format_1_Offset backend (IpTSymLinearSolver.cpp:100-121,357-371,528)
   double t0 = now();
   TripletToCSRConverter conv(1);
   Index nnzc = conv.InitializeConverter(n, nnz, irn.data(), jcn.data());
   double t_conv = now() - t0;
   t0 = now();
   if( iface->InitializeStructure(n, nnzc, conv.IA(), conv.JA()) != SYMSOLVER_SUCCESS ) { fprintf(stderr, "InitializeStructure failed\n"); return 3; }
   double t_factor = 0, t_solve = 0, t_first = 0;
   int neg = -1, status = 0;
   std::vector<Number> x(n);
   for( int it = 0; it < nfactor; ++it )
   {
      Number* vals = iface->GetValuesArrayPtr();
      double tc = now();
      conv.ConvertValues(nnz, a.data(), nnzc, vals);
      x = rhs;
      double t1 = now();
      ESymSolverStatus st = iface->MultiSolve(true, conv.IA(), conv.JA(), 1, x.data(), false, 0);
      double t2 = now();
      status = (int) st;
      neg = iface->NumberOfNegEVals();
      double ts = 0;
      for( int k = 1; k < nsolve; ++k )
      {
         x = rhs;
         double t3 = now();
         iface->MultiSolve(false, conv.IA(), conv.JA(), 1, x.data(), false, 0);
         ts += now() - t3;
      }
      // the first MultiSolve contains symbolic analysis + factor + one solve; later ones factor + solve
      if( it == 0 ) t_first = (t2 - t1) + (t1 - tc);
      else { t_factor += (t2 - tc); t_solve += ts; }
   }
   double resid = 0;   // not recomputed here (bench.py checks x against the GPU solution instead)
   int reps = nfactor > 1 ? nfactor - 1 : 1;
   printf("{\"n\": %d, \"nnz\": %d, \"status\": %d, \"num_neg\": %d, \"convert_init_s\": %.6f, \"first_call_s\": %.6f, "
          "\"factor_plus_first_solve_s\": %.6f, \"extra_solves_s\": %.6f, \"reps\": %d, \"x0\": %.17g, \"xsum\": %.17g}\n",
          n, nnz, status, neg, t_conv, t_first, t_factor / reps, t_solve / reps, reps, x[0], resid);
   return 0;
}
