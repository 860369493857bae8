// ref_kkt_solve.cpp -- TEST / BASELINE INFRASTRUCTURE: times the REFERENCE's own CPU linear-solver path
// (TripletToCSRConverter + PardisoMKLSolverInterface from the unmodified reference, oneMKL PARDISO underneath;
// call sites IpPardisoMKLSolverInterface.cpp:440-715, converter IpTripletToCSRConverter.cpp:46-372) on a KKT
// system given as a binary triplet file.  Used by bench.py's `cpu_baseline` leg ("kind": "reference").
//
// file format (little endian): int32 n, int32 nnz, int32 irn[nnz], int32 jcn[nnz] (1-based), double a[nnz], double rhs[n]
// usage: ref_kkt_solve <file> <nfactor> <nsolve_per_factor> [x_out_file|-] [threads,threads,...]   -> prints one JSON line
//   x_out_file: the solution of the LAST factor+solve is written there (n doubles) -- the parity tests compare the HIP
//   solution and inertia with it; threads: MKL thread counts to time one after the other in ONE process (one symbolic
//   analysis), default = whatever MKL_NUM_THREADS says.
#include "IpIpoptApplication.hpp"
#include "IpPardisoMKLSolverInterface.hpp"
#include "IpTripletToCSRConverter.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>
#include <cmath>

using namespace Ipopt;
extern "C" void MKL_Set_Num_Threads(int);     // oneMKL service routine (libmkl_rt), the same library the reference links for pardiso_
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
   std::vector<int> threads;
   if( argc > 5 ) { char* q = argv[5]; while( *q ) { threads.push_back((int) strtol(q, &q, 10)); if( *q == ',' ) ++q; } }
   if( threads.empty() ) threads.push_back(0);
   int neg = -1, status = 0;
   std::vector<Number> x(n);
   std::vector<double> leg_factor, leg_solve, leg_median;
   double t_first = 0;
   bool first = true;
   for( size_t leg = 0; leg < threads.size(); ++leg )
   {
      if( threads[leg] > 0 ) MKL_Set_Num_Threads(threads[leg]);
      double t_factor = 0, t_solve = 0;
      int timed = 0;
      std::vector<double> steps;       // every timed factor+solves step of the leg (the median is what is reported)
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
         // the very first MultiSolve contains the symbolic analysis; the first of every later leg is a warm-up
         if( first ) { t_first = (t2 - t1) + (t1 - tc); first = false; }
         else if( it == 0 && nfactor > 1 ) { }
         else { t_factor += (t2 - tc); t_solve += ts; ++timed; steps.push_back((t2 - tc) + ts); }
      }
      if( timed == 0 ) timed = 1;
      leg_factor.push_back(t_factor / timed); leg_solve.push_back(t_solve / timed);
      std::sort(steps.begin(), steps.end());
      leg_median.push_back(steps.empty() ? t_factor / timed + t_solve / timed : steps[steps.size() / 2]);
   }
   if( argc > 4 && std::string(argv[4]) != "-" )
   {
      FILE* fo = fopen(argv[4], "wb");
      if( fo ) { fwrite(x.data(), sizeof(Number), n, fo); fclose(fo); }
   }
   size_t best = 0;
   for( size_t leg = 1; leg < threads.size(); ++leg ) if( leg_factor[leg] + leg_solve[leg] < leg_factor[best] + leg_solve[best] ) best = leg;
   double t_factor = leg_factor[best], t_solve = leg_solve[best];
   double resid = 0;   // not recomputed here (bench.py checks x against the GPU solution instead)
   int reps = nfactor > 1 ? nfactor - 1 : 1;
   printf("{\"n\": %d, \"nnz\": %d, \"status\": %d, \"num_neg\": %d, \"convert_init_s\": %.6f, \"first_call_s\": %.6f, "
          "\"factor_plus_first_solve_s\": %.6f, \"extra_solves_s\": %.6f, \"reps\": %d, \"best_threads\": %d, \"legs\": [",
          n, nnz, status, neg, t_conv, t_first, t_factor, t_solve, reps, threads[best]);
   for( size_t leg = 0; leg < threads.size(); ++leg )
      printf("%s{\"threads\": %d, \"factor_plus_first_solve_s\": %.6f, \"extra_solves_s\": %.6f, \"median_step_s\": %.6f}", leg ? ", " : "", threads[leg], leg_factor[leg], leg_solve[leg], leg_median[leg]);
   printf("], \"median_step_s\": %.6f, \"x0\": %.17g, \"xsum\": %.17g}\n", leg_median[best], x[0], resid);
   return 0;
}
