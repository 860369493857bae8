# Recipe that compiles the UNMODIFIED reference (coin-or/Ipopt 3.14.15) from the
# sources where they lie under $(REF)/src into oracle/_ref/ -- without running the
# reference's autotools build.  TEST / ORACLE INFRASTRUCTURE ONLY.
#
#   make -f oracle/ref_build.mk -j8        (from the repo root)
#
# Products (all git-ignored, but they travel to the GPU box with the snapshot):
#   oracle/_ref/libipopt_ref.so   the reference host library, LAPACK+PARDISO from oneMKL
#   oracle/_ref/solve_problem     reference examples/ScalableProblems driver (LukVlE1, MBndryCntrl1, ...)
#                                 linked with tests' own main that can select our backend
#   oracle/_ref/hs071_cpp         reference test/hs071_cpp
# Source list = libipopt_la_SOURCES of $(REF)/src/Makefile.am:85-196 (+ PardisoMKL, :198-200;
# + the !IPOPT_INT64 group, :202-211).  No reference file is copied.
REF     ?= /root/reference
OUT     ?= oracle/_ref
CFG     := oracle/ipopt_cfg
MKLDIR  ?= /opt/conda/lib
CXX     ?= g++
CXXFLAGS_REF := -O2 -DNDEBUG -fPIC -DHAVE_CONFIG_H -DIPOPTLIB_BUILD -std=c++11 -w
INCS := -I$(CFG) -I$(REF)/src/Common -I$(REF)/src/LinAlg -I$(REF)/src/LinAlg/TMatrices \
        -I$(REF)/src/Algorithm -I$(REF)/src/Algorithm/LinearSolvers -I$(REF)/src/Algorithm/Inexact \
        -I$(REF)/src/Interfaces -I$(REF)/src/contrib/CGPenalty

SRCS_CPP := $(wildcard $(REF)/src/Common/*.cpp) $(wildcard $(REF)/src/LinAlg/*.cpp) \
            $(wildcard $(REF)/src/LinAlg/TMatrices/*.cpp) $(wildcard $(REF)/src/Algorithm/*.cpp) \
            $(wildcard $(REF)/src/contrib/CGPenalty/*.cpp) \
            $(addprefix $(REF)/src/Interfaces/, IpInterfacesRegOp.cpp IpIpoptApplication.cpp IpSolveStatistics.cpp \
               IpStdCInterface.cpp IpStdInterfaceTNLP.cpp IpTNLP.cpp IpTNLPAdapter.cpp IpTNLPReducer.cpp) \
            $(addprefix $(REF)/src/Algorithm/LinearSolvers/, IpLinearSolversRegOp.cpp IpSlackBasedTSymScalingMethod.cpp \
               IpTripletToCSRConverter.cpp IpTSymDependencyDetector.cpp IpTSymLinearSolver.cpp \
               IpPardisoMKLSolverInterface.cpp IpMc19TSymScalingMethod.cpp IpMa27TSolverInterface.cpp \
               IpMa57TSolverInterface.cpp IpMa77SolverInterface.cpp IpMa86SolverInterface.cpp \
               IpMa97SolverInterface.cpp IpPardisoSolverInterface.cpp)
SRCS_C   := $(REF)/src/Algorithm/LinearSolvers/IpLinearSolvers.c $(REF)/src/Interfaces/IpStdFInterface.c
OBJS := $(patsubst $(REF)/src/%.cpp,$(OUT)/obj/%.o,$(SRCS_CPP)) $(patsubst $(REF)/src/%.c,$(OUT)/obj/%.o,$(SRCS_C))

# MKL is isolated behind symlinks so that conda's older libstdc++ is never on the link path
MKLLINK := -L$(OUT)/mkl -lmkl_rt -Wl,-rpath,'$$ORIGIN/mkl' -ldl -lm -lpthread

all: $(OUT)/libipopt_ref.so $(OUT)/hs071_cpp $(OUT)/scalable.a $(OUT)/ref_driver $(OUT)/ref_kkt_solve $(OUT)/libmi355x_ipopt.so $(OUT)/ipopt_mi355x_driver \
     $(OUT)/libipopt_ref_mi355x.so $(OUT)/ipopt_patched_driver

$(OUT)/mkl/.stamp:
	mkdir -p $(OUT)/mkl
	for f in $(MKLDIR)/libmkl_* $(MKLDIR)/libiomp5.so; do ln -sf $$f $(OUT)/mkl/; done
	touch $@

$(OUT)/obj/%.o: $(REF)/src/%.cpp
	@mkdir -p $(dir $@)
	@$(CXX) $(CXXFLAGS_REF) $(INCS) -c $< -o $@
$(OUT)/obj/%.o: $(REF)/src/%.c
	@mkdir -p $(dir $@)
	@gcc -O2 -fPIC -DHAVE_CONFIG_H -DIPOPTLIB_BUILD -w $(INCS) -c $< -o $@

$(OUT)/libipopt_ref.so: $(OBJS) $(OUT)/mkl/.stamp
	@$(CXX) -shared -o $@ $(OBJS) $(MKLLINK)

# public-side include path for code compiled against the reference headers
PUBINC := -I$(CFG)/public $(INCS)
$(CFG)/public/.stamp:
	mkdir -p $(CFG)/public
	touch $@

# reference test/hs071_cpp (3 files, compiled in place)
$(OUT)/hs071_cpp: $(OUT)/libipopt_ref.so
	$(CXX) -O2 -DHAVE_CONFIG_H -std=c++11 -w $(INCS) -I$(REF)/examples/hs071_cpp \
	  $(REF)/examples/hs071_cpp/hs071_main.cpp $(REF)/examples/hs071_cpp/hs071_nlp.cpp \
	  -o $@ -L$(OUT) -lipopt_ref -Wl,-rpath,'$$ORIGIN' $(MKLLINK)

# reference examples/ScalableProblems: every problem class, archived (their own main is NOT used;
# tests/support/ipopt_driver.cpp provides a main that can select the MI355X backend)
SCAL_SRCS := $(filter-out %/solve_problem.cpp,$(wildcard $(REF)/examples/ScalableProblems/*.cpp))
SCAL_OBJS := $(patsubst $(REF)/examples/ScalableProblems/%.cpp,$(OUT)/obj/scal/%.o,$(SCAL_SRCS))
$(OUT)/obj/scal/%.o: $(REF)/examples/ScalableProblems/%.cpp
	@mkdir -p $(dir $@)
	@$(CXX) -O2 -fPIC -DHAVE_CONFIG_H -std=c++11 -w $(INCS) -c $< -o $@
$(OUT)/scalable.a: $(SCAL_OBJS)
	@ar rcs $@ $(SCAL_OBJS)

# --- the product's Ipopt adapter (B1), compiled against the reference headers where they lie.  The
#     adapter SOURCE is product code (ipopt_amd/csrc/ipopt_adapter); only its build needs the reference. ---
KKTLIB := ipopt_amd/lib
ADAPTER_SRC := ipopt_amd/csrc/ipopt_adapter/IpMi355xCommBootstrap.cpp ipopt_amd/csrc/ipopt_adapter/IpMi355xSolverInterface.cpp ipopt_amd/csrc/ipopt_adapter/IpMi355xAugSystemSolver.cpp ipopt_amd/csrc/ipopt_adapter/IpMi355xPDSystemSolver.cpp
# route B2's struct layouts against the reference header: static_asserts, member by member -- the adapter build fails if either side drifts
$(OUT)/ma97_layout_check.o: oracle/ma97_layout_check.cpp include/mi355x_ma97.h $(REF)/src/Algorithm/LinearSolvers/hsl_ma97d.h
	@mkdir -p $(OUT)
	$(CXX) -std=c++11 -Wall -Iinclude -I$(REF)/src/Algorithm/LinearSolvers -c $< -o $@
$(OUT)/libmi355x_ipopt.so: $(ADAPTER_SRC) $(wildcard ipopt_amd/csrc/ipopt_adapter/*.hpp) include/mi355x_kkt.h $(OUT)/libipopt_ref.so $(OUT)/ma97_layout_check.o
	$(CXX) -O2 -fPIC -shared -DHAVE_CONFIG_H -std=c++11 -w $(INCS) -Iinclude -Iipopt_amd/csrc/ipopt_adapter $(ADAPTER_SRC) -o $@ \
	  -L$(OUT) -lipopt_ref -L$(KKTLIB) -lmi355x_kkt -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,'$$ORIGIN/../../ipopt_amd/lib'

DRV_INCS := $(INCS) -I$(REF)/examples/hs071_cpp -I$(REF)/examples/ScalableProblems -Iinclude -Iipopt_amd/csrc/ipopt_adapter
# CPU-only driver (reference + MKL PARDISO): the oracle / golden-fixture generator / CPU baseline
$(OUT)/ref_driver: oracle/ref_driver.cpp $(OUT)/libipopt_ref.so $(OUT)/scalable.a
	$(CXX) -O2 -DHAVE_CONFIG_H -std=c++11 -w $(DRV_INCS) $< $(REF)/examples/hs071_cpp/hs071_nlp.cpp $(OUT)/scalable.a -o $@ \
	  -L$(OUT) -lipopt_ref -Wl,-rpath,'$$ORIGIN' $(MKLLINK)
# stand-alone timing of the reference's CPU linear-solver path on one KKT system (bench.py cpu_baseline)
$(OUT)/ref_kkt_solve: oracle/ref_kkt_solve.cpp $(OUT)/libipopt_ref.so
	$(CXX) -O2 -DHAVE_CONFIG_H -std=c++11 -w $(INCS) $< -o $@ -L$(OUT) -lipopt_ref -Wl,-rpath,'$$ORIGIN' $(MKLLINK)
# same driver with the MI355X backend linked in (end-to-end Ipopt runs on the GPU box)
$(OUT)/ipopt_mi355x_driver: oracle/ref_driver.cpp $(OUT)/libipopt_ref.so $(OUT)/scalable.a $(OUT)/libmi355x_ipopt.so
	$(CXX) -O2 -DHAVE_CONFIG_H -DWITH_MI355X -std=c++11 -w $(DRV_INCS) $< $(REF)/examples/hs071_cpp/hs071_nlp.cpp $(OUT)/scalable.a -o $@ \
	  -L$(OUT) -lipopt_ref -lmi355x_ipopt -L$(KKTLIB) -lmi355x_kkt -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,'$$ORIGIN/../../ipopt_amd/lib' $(MKLLINK)

# --- route B1': the reference WITH the `linear_solver=mi355x` patch a maintainer would carry (oracle/patches/linear_solver_mi355x.patch:
#     factory arms in IpAlgBuilder.cpp:427-526 (mi355x), :568-600 and :644-664 (mi355x-device: AugSystemSolver + PDSystemSolver), option
#     registration in IpLinearSolversRegOp.cpp:84-90).  The two patched
#     translation units are produced in the build directory, compiled and removed again; everything else is the unmodified objects. ---
PATCH   := oracle/patches/linear_solver_mi355x.patch
PSRC    := Algorithm/IpAlgBuilder.cpp Algorithm/LinearSolvers/IpLinearSolversRegOp.cpp Interfaces/IpTNLPAdapter.cpp
POBJS   := $(OUT)/obj_patched/IpAlgBuilder.o $(OUT)/obj_patched/IpLinearSolversRegOp.o $(OUT)/obj_patched/IpTNLPAdapter.o \
           $(OUT)/obj_patched/IpMi355xSolverInterface.o $(OUT)/obj_patched/IpMi355xAugSystemSolver.o $(OUT)/obj_patched/IpMi355xPDSystemSolver.o $(OUT)/obj_patched/IpMi355xCommBootstrap.o
$(OUT)/obj_patched/.stamp: $(PATCH) $(wildcard ipopt_amd/csrc/ipopt_adapter/*.cpp) $(wildcard ipopt_amd/csrc/ipopt_adapter/*.hpp) include/mi355x_kkt.h
	rm -rf $(OUT)/obj_patched $(OUT)/patched_src; mkdir -p $(OUT)/obj_patched $(OUT)/patched_src/src/Algorithm/LinearSolvers $(OUT)/patched_src/src/Interfaces
	for f in $(PSRC); do cp $(REF)/src/$$f $(OUT)/patched_src/src/$$f; done
	cd $(OUT)/patched_src && patch -p1 -s < $(CURDIR)/$(PATCH)
	for f in $(PSRC); do $(CXX) $(CXXFLAGS_REF) -DIPOPT_HAS_MI355X $(INCS) -Iinclude -Iipopt_amd/csrc/ipopt_adapter -c $(OUT)/patched_src/src/$$f -o $(OUT)/obj_patched/`basename $$f .cpp`.o || exit 1; done
	$(CXX) $(CXXFLAGS_REF) $(INCS) -Iinclude -Iipopt_amd/csrc/ipopt_adapter -c ipopt_amd/csrc/ipopt_adapter/IpMi355xCommBootstrap.cpp -o $(OUT)/obj_patched/IpMi355xCommBootstrap.o
	$(CXX) $(CXXFLAGS_REF) $(INCS) -Iinclude -Iipopt_amd/csrc/ipopt_adapter -c ipopt_amd/csrc/ipopt_adapter/IpMi355xSolverInterface.cpp -o $(OUT)/obj_patched/IpMi355xSolverInterface.o
	$(CXX) $(CXXFLAGS_REF) $(INCS) -Iinclude -Iipopt_amd/csrc/ipopt_adapter -c ipopt_amd/csrc/ipopt_adapter/IpMi355xAugSystemSolver.cpp -o $(OUT)/obj_patched/IpMi355xAugSystemSolver.o
	$(CXX) $(CXXFLAGS_REF) $(INCS) -Iinclude -Iipopt_amd/csrc/ipopt_adapter -c ipopt_amd/csrc/ipopt_adapter/IpMi355xPDSystemSolver.cpp -o $(OUT)/obj_patched/IpMi355xPDSystemSolver.o
	rm -rf $(OUT)/patched_src
	touch $@
$(OUT)/libipopt_ref_mi355x.so: $(OBJS) $(OUT)/obj_patched/.stamp $(OUT)/mkl/.stamp
	@$(CXX) -shared -o $@ $(filter-out %/Algorithm/IpAlgBuilder.o %/Algorithm/LinearSolvers/IpLinearSolversRegOp.o %/Interfaces/IpTNLPAdapter.o,$(OBJS)) $(POBJS) \
	  -L$(KKTLIB) -lmi355x_kkt -Wl,-rpath,'$$ORIGIN/../../ipopt_amd/lib' $(MKLLINK)
# the same driver against the patched library: `--solver stock --set linear_solver mi355x`
$(OUT)/ipopt_patched_driver: oracle/ref_driver.cpp $(OUT)/libipopt_ref_mi355x.so $(OUT)/scalable.a
	$(CXX) -O2 -DHAVE_CONFIG_H -std=c++11 -w $(DRV_INCS) $< $(REF)/examples/hs071_cpp/hs071_nlp.cpp $(OUT)/scalable.a -o $@ \
	  -L$(OUT) -lipopt_ref_mi355x -L$(KKTLIB) -lmi355x_kkt -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,'$$ORIGIN/../../ipopt_amd/lib' $(MKLLINK)

clean:
	rm -rf $(OUT)
.PHONY: all clean
