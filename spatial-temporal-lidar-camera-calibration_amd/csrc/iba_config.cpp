// The reference's RUN CONFIGURATION -> the C-ABI's structs, host only (no HIP, no yaml-cpp): what main() of iba_global /
// iba_func / iba_local reads with yaml-cpp (iba_global.cpp:412-471, iba_func.cpp:356-406, iba_local.cpp:325-378) from files
// like config/calib/00/iba_calib_global.yml — three top-level maps `io`, `orb`, `runtime` of scalars, inline comments and flow
// sequences ([a, b, c]) behind an OpenCV-style "%YAML:1.0" / "---" head. That subset is parsed here; anything else in a file is
// an error that names the line (no silent defaults: the reference's `.as<T>()` throws on a missing key too).
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>

#include "../../include/iba_mi355x.h"

namespace {
thread_local std::string g_cfg_err;
iba_status cfail(iba_status s, const std::string& m) { g_cfg_err = m; return s; }

std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char)s[a])) ++a;
    while (b > a && std::isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}
// a YAML comment starts at '#' at the start of a value or after white space (outside quotes)
std::string strip_comment(const std::string& s) {
    bool sq = false, dq = false;
    for (size_t i = 0; i < s.size(); ++i) {
        const char c = s[i];
        if (c == '\'' && !dq) sq = !sq;
        else if (c == '"' && !sq) dq = !dq;
        else if (c == '#' && !sq && !dq && (i == 0 || std::isspace((unsigned char)s[i - 1]))) return s.substr(0, i);
    }
    return s;
}
std::string unquote(const std::string& s) {
    if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) return s.substr(1, s.size() - 2);
    return s;
}
}  // namespace

struct iba_run_config {
    std::map<std::string, std::string> kv;          // "section.key" -> raw scalar text (comments stripped, unquoted) or "[a, b]" for a sequence
    std::string file;
    // strings handed out through iba_dataset_paths live here
    std::string frame_id_file, lidar_pose_file, pointcloud_dir, keyframe_dir, map_file;
    mutable std::map<std::string, std::string> joined;

    const std::string* find(const std::string& k) const { auto it = kv.find(k); return it == kv.end() ? nullptr : &it->second; }
};

namespace {

bool get_str(const iba_run_config* c, const char* key, std::string& out, std::string& err) {
    const std::string* v = c->find(key);
    if (!v) { err = c->file + ": key '" + key + "' is missing"; return false; }
    out = *v;
    return true;
}
bool parse_double(const std::string& t, double& out) {
    if (t.empty()) return false;
    std::string s = t;
    // YAML 1.1 floats as yaml-cpp accepts them: ".5", "1.0E-6", ".inf", "-.inf", ".nan"
    if (s == ".inf" || s == "+.inf" || s == ".Inf" || s == ".INF") { out = INFINITY; return true; }
    if (s == "-.inf" || s == "-.Inf" || s == "-.INF") { out = -INFINITY; return true; }
    if (s == ".nan" || s == ".NaN" || s == ".NAN") { out = NAN; return true; }
    char* end = nullptr;
    out = std::strtod(s.c_str(), &end);
    return end && *end == '\0' && end != s.c_str();
}
bool get_double(const iba_run_config* c, const char* key, double& out, std::string& err) {
    std::string s;
    if (!get_str(c, key, s, err)) return false;
    if (!parse_double(s, out)) { err = c->file + ": '" + key + ": " + s + "' is not a number"; return false; }
    return true;
}
bool get_int(const iba_run_config* c, const char* key, int32_t& out, std::string& err) {
    std::string s;
    if (!get_str(c, key, s, err)) return false;
    char* end = nullptr;
    const long v = std::strtol(s.c_str(), &end, 10);
    if (!end || *end != '\0' || end == s.c_str()) { err = c->file + ": '" + key + ": " + s + "' is not an integer"; return false; }
    out = (int32_t)v;
    return true;
}
bool get_bool(const iba_run_config* c, const char* key, int32_t& out, std::string& err) {
    std::string s;
    if (!get_str(c, key, s, err)) return false;
    // yaml-cpp's bool conversion: y / yes / true / on and n / no / false / off in lower, UPPER or Capitalised form
    std::string l; for (char ch : s) l += (char)std::tolower((unsigned char)ch);
    if (l == "true" || l == "yes" || l == "y" || l == "on") { out = 1; return true; }
    if (l == "false" || l == "no" || l == "n" || l == "off") { out = 0; return true; }
    err = c->file + ": '" + key + ": " + s + "' is not a boolean";
    return false;
}
bool get_vec(const iba_run_config* c, const char* key, std::vector<double>& out, std::string& err) {
    std::string s;
    if (!get_str(c, key, s, err)) return false;
    if (s.size() < 2 || s.front() != '[' || s.back() != ']') { err = c->file + ": '" + key + "' is not a flow sequence [..]"; return false; }
    out.clear();
    std::string body = s.substr(1, s.size() - 2), tok;
    for (size_t i = 0; i <= body.size(); ++i) {
        if (i == body.size() || body[i] == ',') {
            tok = trim(tok);
            if (!tok.empty()) { double v; if (!parse_double(tok, v)) { err = c->file + ": '" + key + "' holds '" + tok + "', not a number"; return false; } out.push_back(v); }
            else if (i != body.size() || !out.empty()) { if (!(i == body.size() && trim(body).empty())) { err = c->file + ": '" + key + "' has an empty entry"; return false; } }
            tok.clear();
        } else tok += body[i];
    }
    return true;
}
}  // namespace

extern "C" {

const char* iba_run_config_last_error(void) { return g_cfg_err.c_str(); }

iba_status iba_run_config_load(const char* yaml_file, iba_run_config** out) {
    if (!yaml_file || !out) return cfail(IBA_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    std::ifstream f(yaml_file);
    if (!f) return cfail(IBA_ERR_IO, std::string(yaml_file) + ": cannot open");
    iba_run_config* c = new iba_run_config;
    c->file = yaml_file;
    std::string line, section;
    int ln = 0;
    auto bad = [&](const std::string& what) { const std::string m = std::string(yaml_file) + ":" + std::to_string(ln) + ": " + what; delete c; return cfail(IBA_ERR_IO, m); };
    while (std::getline(f, line)) {
        ++ln;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.find('\t') != std::string::npos && trim(strip_comment(line)).size()) { /* tabs are not indentation in YAML; tolerate them inside values only */ }
        const std::string body = strip_comment(line);
        const std::string t = trim(body);
        if (t.empty()) continue;
        if (t[0] == '%' || t == "---" || t == "...") continue;   // the "%YAML:1.0" directive of OpenCV's dialect, document markers
        size_t indent = 0;
        while (indent < body.size() && body[indent] == ' ') ++indent;
        // key: value — the key ends at the first ':' that is followed by white space or the end of the line
        size_t colon = std::string::npos;
        for (size_t i = 0; i < t.size(); ++i) if (t[i] == ':' && (i + 1 == t.size() || std::isspace((unsigned char)t[i + 1]))) { colon = i; break; }
        if (colon == std::string::npos) return bad("expected 'key: value', got '" + t + "'");
        const std::string key = trim(t.substr(0, colon));
        std::string val = trim(t.substr(colon + 1));
        if (key.empty()) return bad("empty key");
        if (indent == 0) {
            if (!val.empty()) { c->kv[key] = unquote(val); section.clear(); }   // a top-level scalar (none in the reference's files; kept)
            else section = key;
            continue;
        }
        if (section.empty()) return bad("an indented entry outside a section");
        if (val.empty()) return bad("nested maps / block sequences are not part of the reference's run configuration ('" + key + "')");
        if (val[0] == '[') {   // a flow sequence, possibly continued on the following lines
            while (val.find(']') == std::string::npos) {
                std::string more;
                if (!std::getline(f, more)) return bad("unterminated flow sequence '" + key + "'");
                ++ln;
                val += " " + trim(strip_comment(more));
            }
            val = trim(val);
            if (val.back() != ']') return bad("text after the flow sequence '" + key + "'");
        } else if (val[0] == '{' || val[0] == '&' || val[0] == '*' || val[0] == '|' || val[0] == '>') return bad("YAML construct not used by the reference's run configuration: '" + val + "'");
        else val = unquote(val);
        const std::string full = section + "." + key;
        if (c->kv.count(full)) return bad("duplicate key '" + full + "'");
        c->kv[full] = val;
    }
    *out = c;
    return IBA_OK;
}

void iba_run_config_free(iba_run_config* c) { delete c; }

const char* iba_run_config_get(const iba_run_config* c, const char* dotted_key) {
    if (!c || !dotted_key) return nullptr;
    const std::string* v = c->find(dotted_key);
    return v ? v->c_str() : nullptr;
}

const char* iba_run_config_path(const iba_run_config* c, const char* io_key) {
    if (!c || !io_key) return nullptr;
    const std::string* base = c->find("io.BaseDir");
    const std::string* v = c->find(std::string("io.") + io_key);
    if (!base || !v) return nullptr;
    std::string& slot = c->joined[io_key];
    slot = *base;
    if (!slot.empty() && slot.back() != '/') slot += '/';   // checkpath(base_dir) (kitti_tools.h:18-21)
    slot += *v;          // `base_dir + io_config[...]` — plain concatenation, as the reference does (iba_global.cpp:425-430)
    return slot.c_str();
}

iba_status iba_run_config_params(const iba_run_config* c, int32_t local_stage, iba_params* p) {
    if (!c || !p) return cfail(IBA_ERR_INVALID_ARG, "null argument");
    iba_default_params(p);
    std::string err;
    std::vector<double> w;
    int32_t b = 0;
    bool ok = get_double(c, "runtime.max_pixel_dist", p->max_pixel_dist, err);
    if (!local_stage) {   // IBAGlobalParams: iba_global.cpp:436-459 (= iba_func.cpp:381-397)
        ok = ok && get_double(c, "runtime.corr_3d_2d_threshold", p->corr_3d_2d_threshold, err) && get_double(c, "runtime.corr_3d_3d_threshold", p->corr_3d_3d_threshold, err)
             && get_int(c, "runtime.norm_max_pts", p->norm_max_pts, err) && get_int(c, "runtime.norm_min_pts", p->norm_min_pts, err)
             && get_double(c, "runtime.norm_radius", p->norm_radius, err) && get_double(c, "runtime.norm_reg_threshold", p->norm_reg_threshold, err)
             && get_double(c, "runtime.min_diff_dist", p->min_diff_dist, err) && get_vec(c, "runtime.err_weight", w, err) && get_bool(c, "runtime.use_plane", b, err);
        if (ok && w.size() != 2) { ok = false; err = c->file + ": runtime.err_weight must have 2 entries"; }
        if (ok) { p->err_weight[0] = w[0]; p->err_weight[1] = w[1]; p->use_plane = b; }
    } else {              // IBALocalParams: iba_local.cpp:358-377
        ok = ok && get_double(c, "runtime.neigh_radius", p->neigh_radius, err) && get_int(c, "runtime.neigh_max_pts", p->neigh_max_pts, err)
             && get_double(c, "runtime.min_diff_dist", p->local_min_diff_dist, err) && get_double(c, "runtime.norm_reg_threshold", p->local_norm_reg_threshold, err)
             && get_double(c, "runtime.robust_kernel_delta", p->robust_kernel_delta, err);
    }
    if (!ok) return cfail(IBA_ERR_IO, err);
    return IBA_OK;
}

iba_status iba_run_config_paths(iba_run_config* c, int32_t local_stage, iba_dataset_paths* out) {
    if (!c || !out) return cfail(IBA_ERR_INVALID_ARG, "null argument");
    std::string err, base, vo_id, lo;
    bool ok = get_str(c, "io.BaseDir", base, err) && get_str(c, "io.VOIdFile", vo_id, err) && get_str(c, "io.LOFile", lo, err)
              && get_str(c, "io.PointCloudDir", c->pointcloud_dir, err) && get_str(c, "orb.KeyFrameDir", c->keyframe_dir, err) && get_str(c, "orb.MapFile", c->map_file, err);
    int32_t skip = 1, posx = 0, nbest = 0, minw = 0;
    // the key's spelling differs between the stages: iba_global / iba_func read io["PointCloudskip"] (iba_global.cpp:456), iba_local
    // io["PointCloudSkip"] (iba_local.cpp:372). Each stage asks for its own spelling first and accepts the other (the reference ships
    // only the global stage's yml)
    {
        std::string e1;
        const char* first = local_stage ? "io.PointCloudSkip" : "io.PointCloudskip";
        const char* second = local_stage ? "io.PointCloudskip" : "io.PointCloudSkip";
        if (ok && !get_int(c, first, skip, e1) && !get_int(c, second, skip, err)) { ok = false; err = e1; }
    }
    ok = ok && get_bool(c, "io.PointCloudOnlyPositiveX", posx, err)
         && get_int(c, "runtime.num_best_covis", nbest, err) && get_int(c, "runtime.min_covis_weight", minw, err);
    if (!ok) return cfail(IBA_ERR_IO, err);
    if (!base.empty() && base.back() != '/') base += '/';              // checkpath(base_dir) (kitti_tools.h:18-21; iba_global.cpp:421)
    c->frame_id_file = base + vo_id; c->lidar_pose_file = base + lo;   // iba_global.cpp:426, 428
    // checkpath() appends the missing '/' to a directory: the reference concatenates directory + file name
    if (!c->pointcloud_dir.empty() && c->pointcloud_dir.back() != '/') c->pointcloud_dir += '/';
    if (!c->keyframe_dir.empty() && c->keyframe_dir.back() != '/') c->keyframe_dir += '/';
    out->frame_id_file = c->frame_id_file.c_str(); out->lidar_pose_file = c->lidar_pose_file.c_str(); out->pointcloud_dir = c->pointcloud_dir.c_str();
    out->keyframe_dir = c->keyframe_dir.c_str(); out->map_file = c->map_file.c_str();
    // iba_global / iba_func read the two point-cloud flags into their params and then call readPointCloud WITHOUT them
    // (iba_global.cpp:450-451 vs :494); iba_local passes them (iba_local.cpp:394)
    out->pointcloud_skip = local_stage ? skip : 1; out->only_positive_x = local_stage ? posx : 0;
    out->num_best_covis = nbest; out->min_covis_weight = minw;
    return IBA_OK;
}

iba_status iba_run_config_mads(const iba_run_config* c, const double* x0, iba_mads_options* o) {
    if (!c || !x0 || !o) return cfail(IBA_ERR_INVALID_ARG, "null argument");
    iba_default_mads_options(x0, o);
    std::string err;
    std::vector<double> lb, ub, fr;
    int32_t vns = 1;
    bool ok = get_vec(c, "runtime.lb", lb, err) && get_vec(c, "runtime.ub", ub, err) && get_vec(c, "runtime.init_frame", fr, err) && get_double(c, "runtime.min_mesh", o->min_mesh, err)
              && get_int(c, "runtime.max_bbeval", o->max_bb_eval, err) && get_double(c, "runtime.he_threshold", o->he_threshold, err) && get_double(c, "runtime.valid_rate", o->valid_rate, err)
              && get_int(c, "runtime.seed", o->seed, err) && get_bool(c, "runtime.use_vns", vns, err);
    if (ok && (lb.size() != 7 || ub.size() != 7 || fr.size() != 7)) { ok = false; err = c->file + ": runtime.lb / ub / init_frame must have 7 entries"; }
    if (!ok) return cfail(IBA_ERR_IO, err);
    for (int i = 0; i < 7; ++i) { o->lb[i] = x0[i] + lb[i]; o->ub[i] = x0[i] + ub[i]; o->init_frame[i] = fr[i]; }   // iba_global.cpp:530-533
    if (!vns) o->vns_max_idle = 0;
    return IBA_OK;
}

}  // extern "C"
