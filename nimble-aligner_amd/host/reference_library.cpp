// reference_library.cpp -- library JSON -> (AlignFilterConfig, Reference); utils.rs helpers.
// Mirrors src/reference_library.rs:20-226 and src/utils.rs:7-119 of the reference, including the
// panic messages its tests match on (reference_library.rs:328-353, utils.rs:162-168).
#include <cmath>
#include <cstdio>
#include <fstream>
#include <sstream>

#include "json.hpp"
#include "nimble_host.hpp"

namespace nimble {

namespace align {
const char *to_string(FilterReason r) {
  switch (r) {
    case FilterReason::ScoreBelowThreshold: return "Score Below Threshold";
    case FilterReason::DiscardedMultipleMatch: return "Discarded Multiple Match";
    case FilterReason::DiscardedNonzeroMismatch: return "Discarded Nonzero Mismatch";
    case FilterReason::NoMatch: return "No Match";
    case FilterReason::NoMatchAndScoreBelowThreshold: return "No Match and Score Below Threshold";
    case FilterReason::DifferentFilterReasons: return "Different Filter Reasons";
    case FilterReason::NotMatchingPair: return "Required Valid Pair Not Matching";
    case FilterReason::ForceIntersectFailure: return "Force Intersect Failure";
    case FilterReason::ShortRead: return "Short Read";
    case FilterReason::MaxHitsExceeded: return "Max Hits Exceeded";
    case FilterReason::HighEntropy: return "Low Entropy";
    case FilterReason::SuccessfulMatch: return "Successful Match";
    case FilterReason::StrandWasWrong: return "Strandedness Filtered";
    case FilterReason::TriageEmptyEquivalenceClass: return "Equivalence Class Empty After Filters";
    case FilterReason::AboveMismatchThreshold: return "Above Mismatch Threshold";
    case FilterReason::SkippedAlignDueToUnpairedDummy: return "SKipped Align Due To Unpaired Dummy Read";
    case FilterReason::None: return "None";
  }
  return "None";
}
}  // namespace align

namespace reference_library {

const char *const SPECIAL_REVCOMP_FEATURE_NAME_SEPARATOR = "\xC2\xA7";

namespace {

int get_column_index(const std::vector<std::string> &headers, const std::string &h) {
  for (size_t i = 0; i < headers.size(); ++i)
    if (headers[i] == h) return (int)i;
  return -1;
}

std::string json_repr(const json::Value &v) {
  switch (v.kind) {
    case json::Value::Null: return "null";
    case json::Value::Bool: return v.b ? "true" : "false";
    case json::Value::Number: {
      char buf[64];
      if (v.is_int) snprintf(buf, sizeof buf, "%lld", (long long)v.i);
      else snprintf(buf, sizeof buf, "%g", v.num);
      return buf;
    }
    case json::Value::String: return "\"" + v.str + "\"";
    default: return "...";
  }
}

// reference_library.rs:187-207
std::vector<std::string> to_string_vec(const json::Value &v, const std::string &array_name) {
  const auto *arr = v.as_array();
  if (!arr) throw Panic("Error -- could not parse " + array_name + " as array");
  std::vector<std::string> out;
  out.reserve(arr->size());
  for (const auto &e : *arr) {
    const std::string *s = e.as_str();
    if (!s)
      throw Panic("Error -- could not parse " + array_name + " element \"" + json_repr(e) + "\" as a string");
    out.push_back(*s);
  }
  return out;
}

double need_f64(const json::Value &o, const char *key) {
  double d;
  if (!o[key].as_f64(d)) throw Panic(std::string("Error -- could not parse ") + key + " as f64");
  return d;
}
int64_t need_i64(const json::Value &o, const char *key, const char *what) {
  int64_t i;
  if (!o[key].as_i64(i)) throw Panic(std::string("Error -- could not parse ") + key + " as " + what);
  return i;
}
bool need_bool(const json::Value &o, const char *key, const char *label) {
  bool b;
  if (!o[key].as_bool(b)) throw Panic(std::string("Error -- could not parse ") + label + " as boolean");
  return b;
}

}  // namespace

void sanity_check_align_config(const align::AlignFilterConfig &c) {
  if (!(c.score_percent >= 0.0 && c.score_percent <= 1.0)) throw Panic("Error -- score_percent must be between 0 and 1");
  if (c.score_filter < 0) throw Panic("Error -- score_filter must be positive");
  if (!(c.trim_strictness >= 0.0 && c.trim_strictness <= 1.0))
    throw Panic("Error -- trim_strictness must be between 0 and 1");
}

std::pair<align::AlignFilterConfig, Reference> parse_reference_library(const std::string &text,
                                                                       align::LibraryChemistry strand_filter) {
  json::Value v;
  try {
    v = json::parse(text);
  } catch (const std::exception &) {
    throw Panic("Error -- could not parse reference library JSON");
  }
  const json::Value &cfg = v[(size_t)0];
  align::AlignFilterConfig c;
  // field order as in reference_library.rs:27-78 (decides which panic fires first)
  c.score_percent = need_f64(cfg, "score_percent");
  c.score_filter = (int32_t)need_i64(cfg, "score_filter", "int64");
  c.score_threshold = (size_t)need_i64(cfg, "score_threshold", "int64");
  c.num_mismatches = (size_t)need_i64(cfg, "num_mismatches", "int64");
  c.discard_multiple_matches = need_bool(cfg, "discard_multiple_matches", "discard_multiple_mismatches");
  c.require_valid_pair = need_bool(cfg, "require_valid_pair", "require_valid_pair");
  c.discard_multi_hits = (size_t)need_i64(cfg, "discard_multi_hits", "int64");
  int64_t level = need_i64(cfg, "intersect_level", "int64");
  c.max_hits_to_report = (size_t)need_i64(cfg, "max_hits_to_report", "int64");
  switch (level) {
    case 0: c.intersect_level = align::IntersectLevel::NoIntersect; break;
    case 1: c.intersect_level = align::IntersectLevel::IntersectWithFallback; break;
    case 2: c.intersect_level = align::IntersectLevel::ForceIntersect; break;
    default:
      throw Panic("Error -- invalid intersect level in config file. Please choose intersect level 0, 1, or 2.");
  }
  const std::string *group_on_s = cfg["group_on"].as_str();
  if (!group_on_s) throw Panic("Error -- could not parse group_on as string");
  std::string group_on = *group_on_s;
  c.trim_target_length = (size_t)need_i64(cfg, "trim_target_length", "usize");
  c.trim_strictness = need_f64(cfg, "trim_strictness");

  const json::Value &ref = v[(size_t)1];
  std::vector<std::string> headers = to_string_vec(ref["headers"], "headers");
  int name_idx = get_column_index(headers, "sequence_name");
  if (name_idx < 0) throw Panic("Could not find header sequence_name");
  int group_idx;
  if (group_on.empty()) {
    group_idx = name_idx;
  } else {
    group_idx = get_column_index(headers, group_on);
    if (group_idx < 0) throw Panic("Error -- could not find column for group_on " + group_on);
  }
  int seq_idx = get_column_index(headers, "sequence");
  if (seq_idx < 0) throw Panic("Error -- could not find sequences column");
  const auto *cols_json = ref["columns"].as_array();
  if (!cols_json) throw Panic("Error -- could not parse columns as array");
  std::vector<std::vector<std::string>> columns;
  for (const auto &col : *cols_json) columns.push_back(to_string_vec(col, "column"));

  c.reference_genome_size = columns.at((size_t)name_idx).size();
  c.discard_nonzero_mismatch = false;  // reference_library.rs:116
  c.strand_filter = strand_filter;

  // reference_library.rs:128-161: each row, then its reverse-complemented twin named "<name>§rev"
  const size_t num_rows = columns.empty() ? 0 : columns[0].size();
  std::vector<std::vector<std::string>> final_columns(columns.size());
  for (size_t r = 0; r < num_rows; ++r) {
    std::vector<std::string> row, rev;
    for (size_t col = 0; col < columns.size(); ++col) {
      std::string value = columns[col].at(r);
      if ((int)col == seq_idx)
        for (char &ch : value) { if (ch == 'U') ch = 'T'; else if (ch == 'u') ch = 't'; }
      row.push_back(value);
      rev.push_back(value);
    }
    rev[(size_t)name_idx] += std::string(SPECIAL_REVCOMP_FEATURE_NAME_SEPARATOR) + "rev";
    rev[(size_t)seq_idx] = utils::revcomp(rev[(size_t)seq_idx]);
    for (size_t col = 0; col < columns.size(); ++col) {
      final_columns[col].push_back(row[col]);
      final_columns[col].push_back(rev[col]);
    }
  }
  Reference out;
  out.group_on = (size_t)group_idx;
  out.headers = headers;
  out.columns = std::move(final_columns);
  out.sequence_name_idx = (size_t)name_idx;
  out.sequence_idx = (size_t)seq_idx;
  sanity_check_align_config(c);
  return {c, out};
}

std::pair<align::AlignFilterConfig, Reference> get_reference_library(const std::string &path,
                                                                     align::LibraryChemistry strand_filter) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw Panic("Error -- could not read reference library");
  std::stringstream ss;
  ss << f.rdbuf();
  return parse_reference_library(ss.str(), strand_filter);
}

}  // namespace reference_library

namespace utils {

std::pair<std::vector<std::string>, std::vector<std::string>> get_reference_sequence_data(
    const reference_library::Reference &reference) {
  const auto &seqs = reference.columns.at(reference.sequence_idx);
  const auto &names = reference.columns.at(reference.sequence_name_idx);
  std::vector<std::string> out_seqs, out_names;
  for (size_t i = 0; i < seqs.size(); ++i) {
    out_seqs.push_back(seqs[i]);
    if (i >= names.size())
      throw Panic("Error -- could not read library name after JSON parse, corrupted internal state.");
    out_names.push_back(names[i]);
  }
  return {out_seqs, out_names};
}

void write_to_tsv(const std::vector<std::pair<std::vector<std::string>, int32_t>> &results,
                  const std::string &output_path) {
  // append mode; header only when the file is empty (utils.rs:31-42)
  FILE *f = fopen(output_path.c_str(), "ab");
  if (!f) throw Panic("Unable to open file");
  if (fseek(f, 0, SEEK_END) != 0) { fclose(f); throw Panic("Unable to read file metadata"); }
  long size = ftell(f);
  if (size == 0) fputs("feature\tscore\n", f);
  for (const auto &row : results) {
    std::string line;
    for (size_t i = 0; i < row.first.size(); ++i) {
      if (i) line.push_back('\t');
      line += row.first[i];
    }
    line.push_back('\t');
    line += std::to_string(row.second);
    line.push_back('\n');
    if (fwrite(line.data(), 1, line.size(), f) != line.size()) { fclose(f); throw Panic("Unable to write row"); }
  }
  fclose(f);
}

std::string revcomp(const std::string &sequence) {
  std::string out;
  out.reserve(sequence.size());
  for (size_t i = sequence.size(); i-- > 0;) {
    char bp = sequence[i];
    char c;
    switch (bp) {
      case 'a': c = 't'; break;
      case 'c': c = 'g'; break;
      case 't': c = 'a'; break;
      case 'g': c = 'c'; break;
      case 'u': c = 'a'; break;
      case 'A': c = 'T'; break;
      case 'C': c = 'G'; break;
      case 'T': c = 'A'; break;
      case 'G': c = 'C'; break;
      case 'U': c = 'A'; break;
      case 'N': case 'n': c = 'N'; break;
      default: throw Panic(std::string("Input sequence base is not DNA: ") + bp);
    }
    out.push_back(c);
  }
  return out;
}

double shannon_entropy(const std::string &dna) {
  double total = (double)dna.size();
  double f[4] = {0, 0, 0, 0};  // A, T, C, G
  for (char c : dna) {
    if (c == 'A') f[0] += 1.0;
    else if (c == 'T') f[1] += 1.0;
    else if (c == 'C') f[2] += 1.0;
    else if (c == 'G') f[3] += 1.0;
  }
  double e = 0.0;
  for (double x : f) {
    x /= total;
    if (x > 0.0) e += x * std::log2(x);
  }
  return -e;
}

namespace {
// any_ascii + ASCII lower-casing; on this path only U+00A7 ("§" -> "SS") is non-ASCII
std::string lexical_form(const std::string &s) {
  std::string o;
  o.reserve(s.size() + 2);
  for (size_t i = 0; i < s.size(); ++i) {
    unsigned char c = (unsigned char)s[i];
    if (c == 0xC2 && i + 1 < s.size() && (unsigned char)s[i + 1] == 0xA7) { o += "ss"; ++i; continue; }
    if (c >= 'A' && c <= 'Z') c = (unsigned char)(c + 32);
    o.push_back((char)c);
  }
  return o;
}
inline bool dig(char c) { return c >= '0' && c <= '9'; }
}  // namespace

int natural_lexical_cmp(const std::string &s1, const std::string &s2) {
  const std::string a = lexical_form(s1), b = lexical_form(s2);
  size_t i = 0, j = 0;
  while (i < a.size() || j < b.size()) {
    if (i == a.size()) return -1;
    if (j == b.size()) return 1;
    if (dig(a[i]) && dig(b[j])) {
      size_t ie = i, je = j;
      while (ie < a.size() && dig(a[ie])) ++ie;
      while (je < b.size() && dig(b[je])) ++je;
      size_t is = i, js = j;
      while (is + 1 < ie && a[is] == '0') ++is;
      while (js + 1 < je && b[js] == '0') ++js;
      if (ie - is != je - js) return (ie - is) < (je - js) ? -1 : 1;
      int c = a.compare(is, ie - is, b, js, je - js);
      if (c) return c < 0 ? -1 : 1;
      i = ie;
      j = je;
    } else {
      if (a[i] != b[j]) return (unsigned char)a[i] < (unsigned char)b[j] ? -1 : 1;
      ++i;
      ++j;
    }
  }
  int c = s1.compare(s2);
  return c < 0 ? -1 : (c > 0 ? 1 : 0);
}

}  // namespace utils
}  // namespace nimble
