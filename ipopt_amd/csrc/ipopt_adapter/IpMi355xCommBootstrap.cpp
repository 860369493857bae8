// IpMi355xCommBootstrap.cpp -- see the header.
#include "IpMi355xCommBootstrap.hpp"
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <string>
#include <unistd.h>
#include <fcntl.h>
#include <ctime>
#include <sys/stat.h>
#include <sys/types.h>

namespace Ipopt
{

void Mi355xCommBootstrap::ReadOptions(const OptionsList& options, const std::string& prefix, mi355x_kkt_options& kopts)
{
   try
   {
      Index iv;
      std::string sv;
      if( options.GetIntegerValue("mi355x_nranks", iv, prefix) )
      {
         nranks_opt_ = iv;
      }
      if( options.GetIntegerValue("mi355x_rank", iv, prefix) )
      {
         rank_opt_ = iv;
      }
      if( options.GetStringValue("mi355x_comm_file", sv, prefix) )
      {
         comm_file_ = sv;
      }
      if( options.GetStringValue("mi355x_comm", sv, prefix) )
      {
         use_shm_ = sv == "shm";
      }
      if( options.GetStringValue("mi355x_subcube", sv, prefix) )
      {
         kopts.subcube = sv == "yes" ? 1 : 0;
      }
   }
   catch( ... )
   {
      // unregistered options: the environment decides
   }
   // launcher conventions: torchrun (RANK / WORLD_SIZE / LOCAL_RANK), Open MPI (OMPI_COMM_WORLD_*)
   const char* e;
   int nranks = nranks_opt_, rank = rank_opt_;
   if( nranks <= 0 )
   {
      nranks = (e = getenv("WORLD_SIZE")) ? atoi(e) : ((e = getenv("OMPI_COMM_WORLD_SIZE")) ? atoi(e) : 1);
   }
   if( rank < 0 )
   {
      rank = (e = getenv("RANK")) ? atoi(e) : ((e = getenv("OMPI_COMM_WORLD_RANK")) ? atoi(e) : 0);
   }
   kopts.nranks = nranks > 0 ? nranks : 1;
   kopts.rank = rank;
   if( kopts.nranks > 1 && kopts.device < 0 && !use_shm_ )
   {
      // (mi355x_comm shm is for ranks that may SHARE a device: there the device stays what mi355x_device / the current device says)
      kopts.device = (e = getenv("LOCAL_RANK")) ? atoi(e) : ((e = getenv("OMPI_COMM_WORLD_LOCAL_RANK")) ? atoi(e) : rank);
   }
   if( (e = getenv("MI355X_KKT_COMM")) && !strcmp(e, "shm") )
   {
      use_shm_ = true;
   }
   if( comm_file_.empty() && (e = getenv("MI355X_KKT_COMM_FILE")) )
   {
      comm_file_ = e;
   }
   ready_ = false;
}

bool Mi355xCommBootstrap::Wanted(const mi355x_kkt_options& kopts) const
{
   return kopts.nranks > 1 || getenv("MI355X_KKT_FORCE_MULTI") != NULL;
}

// Rendez-vous of the ranks of one job: rank 0 creates the ncclUniqueId and hands it to the others through a small file.
//   * the record is {magic, job tag, generation, id}: the job tag comes from the launcher's environment (MI355X_KKT_JOB_ID, or torchrun's
//     TORCHELASTIC_RUN_ID / MASTER_PORT, SLURM_JOB_ID, or Open MPI's PMIX_NAMESPACE / OMPI_MCA_ess_base_jobid), the generation counts the communicators this process has set up (every rank
//     sets them up in the same order) -- a reader only accepts the record of ITS job and ITS generation, a file left behind by an earlier
//     run or by the previous set-up of the same run is ignored (and, without a launcher tag, so is any file older than this process);
//   * rank 0 unlinks whatever is there first, writes a private temporary (O_EXCL, 0600) and renames it into place; it removes the file
//     again once ncclCommInitRank has returned, i.e. once every rank has read it;
//   * default location: a per-user directory (0700) under $XDG_RUNTIME_DIR or /tmp, not a fixed world-writable name.
namespace
{
struct CommRecord
{
   unsigned int magic, generation;
   unsigned long long job;
   unsigned char id[128];
};
const unsigned int COMM_MAGIC = 0x4b4b4d49u;      // "IMKK"

unsigned long long comm_job_tag()
{
   // (Open MPI's mpirun / PRRTE set none of the first three: OMPI_MCA_ess_base_jobid / PMIX_NAMESPACE identify the job there)
   const char* names[6] = {"MI355X_KKT_JOB_ID", "TORCHELASTIC_RUN_ID", "SLURM_JOB_ID", "PMIX_NAMESPACE", "OMPI_MCA_ess_base_jobid", "MASTER_PORT"};
   for( int q = 0; q < 6; ++q )
   {
      const char* e = getenv(names[q]);
      if( e && *e )
      {
         unsigned long long h = 1469598103934665603ull;      // FNV-1a of "<name>=<value>"
         for( const char* c = names[q]; *c; ++c ) { h = (h ^ (unsigned char) *c) * 1099511628211ull; }
         for( const char* c = e; *c; ++c ) { h = (h ^ (unsigned char) *c) * 1099511628211ull; }
         return h ? h : 1ull;
      }
   }
   return 0ull;      // no launcher tag: readers fall back to "not older than this process"
}

std::string comm_default_path(unsigned long long job)
{
   const char* rt = getenv("XDG_RUNTIME_DIR");
   char buf[64];
   snprintf(buf, sizeof(buf), "/mi355x_kkt_%u", (unsigned) getuid());
   const std::string dir = std::string((rt && *rt) ? rt : "/tmp") + buf;
   (void) mkdir(dir.c_str(), 0700);
   snprintf(buf, sizeof(buf), "/comm_id_%016llx", job);
   return dir + buf;
}
const time_t g_process_start = time(NULL);
}

bool Mi355xCommBootstrap::Setup(mi355x_kkt_handle handle, const mi355x_kkt_options& kopts, const Journalist& jnlst)
{
   CommRecord rec;
   memset(&rec, 0, sizeof(rec));
   const unsigned long long job = comm_job_tag();
   const unsigned int generation = ++generation_;
   const std::string path = comm_file_.empty() ? comm_default_path(job) : comm_file_;
   if( job == 0ull && kopts.nranks > 1 && generation == 1 )
   {
      jnlst.Printf(J_WARNING, J_LINEAR_ALGEBRA, "mi355x: rank %d found no job tag in the environment (MI355X_KKT_JOB_ID, TORCHELASTIC_RUN_ID, SLURM_JOB_ID, "
                     "PMIX_NAMESPACE, OMPI_MCA_ess_base_jobid, MASTER_PORT): the communicator id file %s is only protected by its age; set MI355X_KKT_JOB_ID on every rank\n",
                     kopts.rank, path.c_str());
   }
   if( kopts.rank == 0 )
   {
      if( use_shm_ )
      {
         if( mi355x_kkt_comm_shm_id(rec.id, kopts.nranks) != MI355X_KKT_SUCCESS )
         {
            jnlst.Printf(J_ERROR, J_LINEAR_ALGEBRA, "mi355x: could not create the shared-memory segment of mi355x_comm shm\n");
            return false;
         }
      }
      else if( mi355x_kkt_comm_unique_id(rec.id) != MI355X_KKT_SUCCESS )
      {
         jnlst.Printf(J_ERROR, J_LINEAR_ALGEBRA, "mi355x: could not create the RCCL unique id (librccl.so not loadable?)\n");
         return false;
      }
      if( kopts.nranks > 1 )
      {
         rec.magic = COMM_MAGIC; rec.generation = generation; rec.job = job;
         char sfx[48];
         snprintf(sfx, sizeof(sfx), ".tmp.%ld.%u", (long) getpid(), generation);
         const std::string tmp = path + sfx;
         (void) unlink(path.c_str());                       // whatever an earlier run or set-up left behind
         (void) unlink(tmp.c_str());
         const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL, 0600);
         if( fd < 0 || write(fd, &rec, sizeof(rec)) != (ssize_t) sizeof(rec) )
         {
            jnlst.Printf(J_ERROR, J_LINEAR_ALGEBRA, "mi355x: cannot write %s\n", tmp.c_str());
            if( fd >= 0 ) close(fd);
            if( use_shm_ ) mi355x_kkt_comm_shm_discard(rec.id);
            return false;
         }
         close(fd);
         if( rename(tmp.c_str(), path.c_str()) != 0 )
         {
            (void) unlink(tmp.c_str());
            if( use_shm_ ) mi355x_kkt_comm_shm_discard(rec.id);
            return false;
         }
      }
   }
   else
   {
      bool got = false, aged = false;
      for( int tries = 0; tries < 6000 && !got; ++tries )   // up to 10 minutes: rank 0 may still be in its (longer) start-up
      {
         struct stat sb;
         FILE* f = fopen(path.c_str(), "rb");
         if( f )
         {
            CommRecord in;
            const bool whole = fread(&in, 1, sizeof(in), f) == sizeof(in);
            // without any job tag the only protection against a stale file of an earlier job is its age: written no more than
            // MI355X_KKT_COMM_FRESH_S seconds before this rank loaded the library.  The default is 120 s (30 s rejected valid records of launchers that start their ranks one after the other): rank 0 unlinks what an earlier
            // run left behind and writes its record after its own start, so a record that is older than this rank's start by more than the
            // launcher's stagger is a leftover of a run that died between its write and its join -- accepting it would hang ncclCommInitRank.
            // Launchers with a larger stagger set a job tag (then no age test at all) or the window.
            static const long fresh_s = getenv("MI355X_KKT_COMM_FRESH_S") ? atol(getenv("MI355X_KKT_COMM_FRESH_S")) : 120;
            const bool have_stat = fstat(fileno(f), &sb) == 0;
            const bool fresh = job != 0ull || (have_stat && sb.st_mtime + fresh_s >= g_process_start);
            aged = whole && !fresh;
            if( whole && !fresh && tries % 100 == 0 )
            {
               jnlst.Printf(J_WARNING, J_LINEAR_ALGEBRA, "mi355x: rank %d ignores %s: no job tag in the environment and the file is older than %ld s "
                              "(set MI355X_KKT_JOB_ID on every rank, or MI355X_KKT_COMM_FRESH_S)\n", kopts.rank, path.c_str(), fresh_s);
            }
            fclose(f);
            if( whole && fresh && in.magic == COMM_MAGIC && in.job == job && in.generation == generation )
            {
               rec = in;
               got = true;
            }
         }
         if( !got )
         {
            usleep(100000);
         }
      }
      if( !got )
      {
         jnlst.Printf(J_ERROR, J_LINEAR_ALGEBRA, "mi355x: rank %d never saw generation %u of the communicator id file %s%s\n", kopts.rank, generation, path.c_str(),
                      aged ? " -- a whole record was there but was rejected ONLY for its age: set MI355X_KKT_JOB_ID on every rank (or raise MI355X_KKT_COMM_FRESH_S)" : "");
         return false;
      }
   }
   if( (use_shm_ ? mi355x_kkt_set_comm_shm(handle, rec.id) : mi355x_kkt_set_comm_rccl(handle, rec.id)) != MI355X_KKT_SUCCESS )
   {
      jnlst.Printf(J_ERROR, J_LINEAR_ALGEBRA, "mi355x_kkt_set_comm_%s failed: %s\n", use_shm_ ? "shm" : "rccl", mi355x_kkt_last_error(handle));
      return false;
   }
   if( kopts.rank == 0 && kopts.nranks > 1 )
   {
      (void) unlink(path.c_str());                          // ncclCommInitRank has returned: every rank has read the record
   }
   ready_ = true;
   jnlst.Printf(J_DETAILED, J_LINEAR_ALGEBRA, "MI355X: rank %d of %d joined the %s communicator (device %d)\n", kopts.rank, kopts.nranks, use_shm_ ? "shared-memory" : "RCCL", kopts.device);
   return true;
}


} // namespace Ipopt
