// main.cpp -- `nimble` command line, argv-compatible with the reference's clap schema
// (src/bin/cli.yml:5-49) and control flow of src/bin/main.rs:12-162.  FASTQ inputs run on the GPU path;
// BAM inputs are outside this build's scope (htslib is not part of it) and are rejected explicitly.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "nimble_host.hpp"

using namespace nimble;

namespace {

[[noreturn]] void usage(const char *msg) {
  if (msg) fprintf(stderr, "error: %s\n\n", msg);
  fprintf(stderr,
          "nimble 0.8.0-mi355x\n"
          "Fast, configurable sequence alignment tool on arbitrary reference libraries\n\n"
          "USAGE:\n    nimble [FLAGS] [OPTIONS] --input <input>... --output <output>... --reference <reference>...\n\n"
          "FLAGS:\n    -p, --force_bam_paired\n    -h, --help\n    -V, --version\n\n"
          "OPTIONS:\n"
          "    -i, --input <input>...             .fastq.gz/fastq file(s), or a single .bam file\n"
          "    -c, --cores <NUMBER_OF_CORES>      The number of cores to use during alignment [default: 1]\n"
          "    -o, --output <output>...           Output TSV file name(s)\n"
          "    -r, --reference <reference>...     Reference libraries in nimble .json format\n"
          "    -f, --strand_filter <STRAND_FILTER>  unstranded (default), fiveprime, threeprime, none\n"
          "    -t, --trim <TRIM>                  <TARGET_LENGTH>:<STRICTNESS>, comma-separated, one per library\n"
          "    -d, --device <ORDINAL>[,<ORDINAL>...]  HIP device(s) to run on [default: 0]; several = one rank per device,\n"
          "                                       reads exchanged and counts summed over RCCL (FASTQ input)\n");
  exit(msg ? 1 : 0);
}

bool is_flag(const std::string &a) { return a.size() >= 2 && a[0] == '-' && !(a[1] >= '0' && a[1] <= '9'); }

std::string lower(std::string s) {
  for (char &c : s) c = (char)tolower((unsigned char)c);
  return s;
}

}  // namespace

int main(int argc, char **argv) {
  std::vector<std::string> refs, outs, ins;
  std::string cores = "1", strand = "unstranded", trim;
  bool have_trim = false, have_cores = false, force_bam_paired = false;
  std::vector<int> devices{0};
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto multi = [&](std::vector<std::string> &dst) {
      while (i + 1 < argc && !is_flag(argv[i + 1])) dst.push_back(argv[++i]);
    };
    auto single = [&](std::string &dst) {
      if (i + 1 >= argc) usage(("missing value for " + a).c_str());
      dst = argv[++i];
    };
    if (a == "-r" || a == "--reference") multi(refs);
    else if (a == "-o" || a == "--output") multi(outs);
    else if (a == "-i" || a == "--input") multi(ins);
    else if (a == "-c" || a == "--cores") { single(cores); have_cores = true; }
    else if (a == "-f" || a == "--strand_filter") single(strand);
    else if (a == "-t" || a == "--trim") { single(trim); have_trim = true; }
    else if (a == "-p" || a == "--force_bam_paired") force_bam_paired = true;
    else if (a == "-d" || a == "--device") {
      std::string d;
      single(d);
      devices.clear();
      for (size_t p = 0; p <= d.size();) {
        size_t q = d.find(',', p);
        devices.push_back(atoi(d.substr(p, q == std::string::npos ? std::string::npos : q - p).c_str()));
        if (q == std::string::npos) break;
        p = q + 1;
      }
    }
    else if (a == "-h" || a == "--help") usage(nullptr);
    else if (a == "-V" || a == "--version") { puts("nimble 0.8.0-mi355x"); return 0; }
    else usage(("unexpected argument " + a).c_str());
  }
  if (refs.empty() || outs.empty() || ins.empty()) usage("--reference, --output and --input are required");
  try {
    char *end = nullptr;
    (void)strtoull(cores.c_str(), &end, 10);
    if (!end || *end) throw Panic("Error -- please provide an integer value for the number of cores");
    // the reference spends its cores on alignment (build_index threads, the BAM consumer pool); here they are the
    // host threads of the index build and of the FASTQ parser.  Without -c the host picks (up to 8 parser threads).
    if (have_cores) {
      setenv("NIMBLE_INDEX_THREADS", cores.c_str(), 0);
      setenv("NIMBLE_FASTQ_THREADS", cores.c_str(), 0);
    }
    align::LibraryChemistry chem;
    if (strand == "unstranded") chem = align::LibraryChemistry::Unstranded;
    else if (strand == "fiveprime") chem = align::LibraryChemistry::FivePrime;
    else if (strand == "threeprime") chem = align::LibraryChemistry::ThreePrime;
    else if (strand == "none") chem = align::LibraryChemistry::None;
    else throw Panic("Could not parse strand_filter option.");

    const std::string &first = ins[0];
    size_t slash = first.find_last_of('/');
    std::string fname = slash == std::string::npos ? first : first.substr(slash + 1);
    size_t dot = fname.find_last_of('.');
    std::string ext = (dot == std::string::npos || dot == 0) ? "" : lower(fname.substr(dot + 1));
    bool is_fastq_gz = fname.size() >= 9 && fname.compare(fname.size() - 9, 9, ".fastq.gz") == 0;

    std::vector<std::pair<size_t, double>> trim_pairs;
    if (have_trim) {
      size_t p = 0;
      while (p <= trim.size()) {
        size_t q = trim.find(',', p);
        std::string item = trim.substr(p, q == std::string::npos ? std::string::npos : q - p);
        size_t colon = item.find(':');
        char *e1 = nullptr, *e2 = nullptr;
        std::string ls = item.substr(0, colon), ss = colon == std::string::npos ? "" : item.substr(colon + 1);
        unsigned long long len = strtoull(ls.c_str(), &e1, 10);
        if (ls.empty() || *e1) throw Panic("Invalid length");
        double st = strtod(ss.c_str(), &e2);
        if (ss.empty() || *e2) throw Panic("Invalid strictness");
        trim_pairs.emplace_back((size_t)len, st);
        if (q == std::string::npos) break;
        p = q + 1;
      }
      if (trim_pairs.size() != refs.size())
        throw Panic("The number of trim options does not match the number of reference libraries");
    }

    std::vector<std::unique_ptr<align::PseudoAligner>> indices;
    std::vector<std::vector<std::unique_ptr<align::PseudoAligner>>> rank_indices;  // [library][rank], several devices
    const bool sharded = devices.size() > 1;
    std::vector<reference_library::Reference> references;
    std::vector<align::AlignFilterConfig> configs;
    for (size_t i = 0; i < refs.size(); ++i) {
      printf("Loading and preprocessing reference data for %s\n", refs[i].c_str());
      auto pr = reference_library::get_reference_library(refs[i], chem);
      if (i < trim_pairs.size()) {
        pr.first.trim_target_length = trim_pairs[i].first;
        pr.first.trim_strictness = trim_pairs[i].second;
        printf("Manually setting trim settings for library %s | target length: %zu, strictness: %g\n", refs[i].c_str(),
               pr.first.trim_target_length, pr.first.trim_strictness);
      }
      auto data = utils::get_reference_sequence_data(pr.second);
      if (!sharded) {
        indices.push_back(align::PseudoAligner::build_index(data.first, data.second, devices[0]));
      } else {  // the library's index on every rank's device
        rank_indices.emplace_back();
        for (int dv : devices) rank_indices.back().push_back(align::PseudoAligner::build_index(data.first, data.second, dv));
      }
      references.push_back(std::move(pr.second));
      configs.push_back(pr.first);
    }
    puts("Loading read sequences and aligning");
    if (is_fastq_gz || ext == "fastq") {
      puts("Processing as FASTQ file");
      if (sharded) process::fastq::process_sharded(ins, rank_indices, references, configs, outs, devices);
      else process::fastq::process(ins, indices, references, configs, outs);
    } else if (ext == "bam") {
      puts("Processing as BAM file");
      if (sharded) throw Panic("BAM input runs on one device (-d takes one ordinal)");
      process::bam::process(ins, indices, references, configs, outs, (size_t)strtoull(cores.c_str(), nullptr, 10),
                            force_bam_paired);
    } else {
      throw Panic("Unsupported file format: " + ext);
    }
    puts("Alignment successful, terminating.");
  } catch (const std::exception &e) {
    fprintf(stderr, "thread 'main' panicked: %s\n", e.what());
    return 101;  // exit status of a Rust panic
  }
  return 0;
}
