// tmx_trajopt_json.hpp — ProblemConstructionInfo from the reference's problem-description JSON, for the C++ host layer.
//
// Mirrors trajopt::ProblemConstructionInfo::fromJson / readBasicInfo / readOptInfo / readCosts / readConstraints /
// readInitInfo (trajopt/src/problem_description.cpp:118-308) and the fromJson methods of the term classes this path
// lowers (JointPosTermInfo :1059-1071, JointVelTermInfo :1178-1195, CartPoseTermInfo :823-855, CollisionTermInfo
// :1617-1714), restricted like trajopt_amd/json_io.py: a term, option or back-end the device path does not lower throws
// (never a CPU detour); unknown parameter names throw as json_marshal::ensure_only_members does (:66-79).
// The reference parses with jsoncpp; no JSON library is part of this toolchain, so a minimal recursive-descent reader
// (objects, arrays, strings, numbers, true / false / null) is included here.  Header-only, C++17.
#ifndef TMX_TRAJOPT_JSON_HPP_
#define TMX_TRAJOPT_JSON_HPP_

#include <cctype>
#include <cstdlib>
#include <initializer_list>

#include "tmx_trajopt.hpp"

namespace tmx
{
namespace json
{
struct Value
{
  enum Kind
  {
    NUL,
    BOOL,
    NUMBER,
    STRING,
    ARRAY,
    OBJECT
  };
  Kind kind{ NUL };
  bool b{ false };
  double num{ 0 };
  std::string str;
  std::vector<Value> arr;
  std::vector<std::pair<std::string, Value>> obj;  // insertion order kept (cost / constraint order matters)

  bool isMember(const std::string& k) const
  {
    for (const auto& kv : obj)
      if (kv.first == k)
        return true;
    return false;
  }
  const Value& operator[](const std::string& k) const
  {
    for (const auto& kv : obj)
      if (kv.first == k)
        return kv.second;
    printAndThrow("missing required field \"" + k + "\"");
  }
  double asDouble() const
  {
    if (kind != NUMBER)
      printAndThrow("JSON value is not a number");
    return num;
  }
  int asInt() const { return static_cast<int>(asDouble()); }
  /** Json::Value::asBool(): booleans and numbers convert; a STRING does not ("Value is not convertible to bool") - which is what
      "use_time" : "false" of trajopt_common/data/config/arm_around_table_time.json runs into in the reference (json_marshal.cpp:10-20) */
  bool asBool() const
  {
    if (kind == NUMBER)
      return num != 0.0;
    if (kind != BOOL)
      printAndThrow("expected: bool, got a value that is not convertible to bool");
    return b;
  }
  const std::string& asString() const
  {
    if (kind != STRING)
      printAndThrow("JSON value is not a string");
    return str;
  }
};

class Parser
{
public:
  explicit Parser(const std::string& text) : s_(text) {}
  Value parse()
  {
    Value v = value();
    ws();
    if (i_ != s_.size())
      fail("trailing characters");
    return v;
  }

private:
  [[noreturn]] void fail(const std::string& what) const { printAndThrow("JSON parse error at offset " + std::to_string(i_) + ": " + what); }
  void ws()
  {
    while (i_ < s_.size() && std::isspace(static_cast<unsigned char>(s_[i_])))
      ++i_;
  }
  bool eat(char c)
  {
    ws();
    if (i_ < s_.size() && s_[i_] == c)
    {
      ++i_;
      return true;
    }
    return false;
  }
  Value value()
  {
    ws();
    if (i_ >= s_.size())
      fail("unexpected end");
    const char c = s_[i_];
    Value v;
    if (c == '{')
    {
      ++i_;
      v.kind = Value::OBJECT;
      if (eat('}'))
        return v;
      do
      {
        ws();
        Value k = string();
        if (!eat(':'))
          fail("':' expected");
        v.obj.emplace_back(k.str, value());
      } while (eat(','));
      if (!eat('}'))
        fail("'}' expected");
    }
    else if (c == '[')
    {
      ++i_;
      v.kind = Value::ARRAY;
      if (eat(']'))
        return v;
      do
        v.arr.push_back(value());
      while (eat(','));
      if (!eat(']'))
        fail("']' expected");
    }
    else if (c == '"')
      v = string();
    else if (s_.compare(i_, 4, "true") == 0)
    {
      v.kind = Value::BOOL;
      v.b = true;
      i_ += 4;
    }
    else if (s_.compare(i_, 5, "false") == 0)
    {
      v.kind = Value::BOOL;
      i_ += 5;
    }
    else if (s_.compare(i_, 4, "null") == 0)
      i_ += 4;
    else
    {
      const char* begin = s_.c_str() + i_;
      char* end = nullptr;
      v.num = std::strtod(begin, &end);
      if (end == begin)
        fail("value expected");
      v.kind = Value::NUMBER;
      i_ += static_cast<std::size_t>(end - begin);
    }
    return v;
  }
  Value string()
  {
    if (i_ >= s_.size() || s_[i_] != '"')
      fail("string expected");
    ++i_;
    Value v;
    v.kind = Value::STRING;
    while (i_ < s_.size() && s_[i_] != '"')
    {
      char c = s_[i_++];
      if (c == '\\' && i_ < s_.size())
      {
        const char e = s_[i_++];
        c = (e == 'n') ? '\n' : (e == 't') ? '\t' : (e == 'r') ? '\r' : (e == 'b') ? '\b' : (e == 'f') ? '\f' : e;  // \uXXXX not needed here
      }
      v.str.push_back(c);
    }
    if (i_ >= s_.size())
      fail("unterminated string");
    ++i_;
    return v;
  }
  const std::string& s_;
  std::size_t i_{ 0 };
};

inline Value parse(const std::string& text) { return Parser(text).parse(); }
}  // namespace json

namespace trajopt
{
namespace detail
{
/** json_marshal::ensure_only_members (problem_description.cpp:66-79) */
inline void ensureOnlyMembers(const json::Value& v, std::initializer_list<const char*> allowed, const std::string& what)
{
  for (const auto& kv : v.obj)
  {
    bool ok = false;
    for (const char* a : allowed)
      ok = ok || kv.first == a;
    if (!ok)
      printAndThrow(what + ": illegal field \"" + kv.first + "\"");
  }
}
inline DblVec jsonVec(const json::Value& params, const std::string& key, std::size_t n, const DblVec* dflt)
{
  if (!params.isMember(key))
  {
    if (!dflt)
      printAndThrow("missing required field \"" + key + "\"");
    return *dflt;
  }
  const json::Value& v = params[key];
  DblVec out;
  if (v.kind == json::Value::ARRAY)
    for (const auto& e : v.arr)
      out.push_back(e.asDouble());
  else
    out.push_back(v.asDouble());
  if (out.size() == 1 && n > 1 && key == "coeffs")
    out = DblVec(n, out[0]);  // arm_around_table.json style: "coeffs": [1] broadcast over the joints
  if (out.size() != n)
    printAndThrow("wrong number of values in \"" + key + "\": expected " + std::to_string(n) + " got " + std::to_string(out.size()));
  return out;
}
inline int jsonInt(const json::Value& params, const std::string& key, int dflt) { return params.isMember(key) ? params[key].asInt() : dflt; }
inline double jsonDouble(const json::Value& params, const std::string& key, double dflt)
{
  return params.isMember(key) ? params[key].asDouble() : dflt;
}
inline Transform jsonOffset(const json::Value& p, const std::string& xyz_key, const std::string& wxyz_key)
{
  const DblVec zero3(3, 0.0), unit4 = { 1, 0, 0, 0 };
  const DblVec t = jsonVec(p, xyz_key, 3, &zero3), q = jsonVec(p, wxyz_key, 4, &unit4);
  return Transform::FromQuaternion(q[0], q[1], q[2], q[3], t[0], t[1], t[2]);
}

inline TermInfo::Ptr readTerm(const json::Value& it, bool is_cost, const ProblemConstructionInfo& pci)
{
  const std::string typ = it["type"].asString();
  // readCosts / readConstraints (problem_description.cpp:162-216): a term-level "use_time" sets TT_USE_TIME (and basic_info.use_time,
  // done by the caller)
  const bool term_time = it.isMember("use_time") && it["use_time"].asBool();
  if (!it.isMember("params"))
    printAndThrow(typ + ": missing params");
  const json::Value& p = it["params"];
  const std::string name = it.isMember("name") ? it["name"].asString() : typ;
  const std::size_t D = pci.kin->numJoints();
  const int n_steps = pci.basic_info.n_steps;
  const DblVec ones(D, 1.0), zeros(D, 0.0);
  const TermType tt = term_time ? ((is_cost ? TermType::TT_COST : TermType::TT_CNT) | TermType::TT_USE_TIME) : (is_cost ? TermType::TT_COST : TermType::TT_CNT);
  if (typ == "total_time")
  {
    // TotalTimeTermInfo::fromJson (problem_description.cpp:1839-1850)
    ensureOnlyMembers(p, { "coeff", "limit" }, typ);
    auto t = std::make_shared<TotalTimeTermInfo>();
    t->coeff = jsonDouble(p, "coeff", 1.0);
    t->limit = jsonDouble(p, "limit", 1.0);
    t->name = name;
    t->term_type = tt;
    return t;
  }
  if (typ == "joint_vel")
  {
    ensureOnlyMembers(p, { "coeffs", "first_step", "last_step", "targets", "lower_tols", "upper_tols", "use_time" }, typ);
    auto t = std::make_shared<JointVelTermInfo>();
    t->coeffs = jsonVec(p, "coeffs", D, &ones);
    t->targets = jsonVec(p, "targets", D, nullptr);
    t->upper_tols = jsonVec(p, "upper_tols", D, &zeros);
    t->lower_tols = jsonVec(p, "lower_tols", D, &zeros);
    t->first_step = jsonInt(p, "first_step", 0);
    t->last_step = jsonInt(p, "last_step", n_steps - 1);
    t->name = name;
    t->term_type = tt;
    return t;
  }
  if (typ == "joint_acc" || typ == "joint_jerk")
  {
    // JointAccTermInfo::fromJson / JointJerkTermInfo::fromJson (problem_description.cpp:1374-1391, :1495-1513)
    ensureOnlyMembers(p, { "coeffs", "first_step", "last_step", "targets", "lower_tols", "upper_tols", "use_time" }, typ);
    auto fill = [&](auto t) {
      t->coeffs = jsonVec(p, "coeffs", D, &ones);
      t->targets = jsonVec(p, "targets", D, nullptr);
      t->upper_tols = jsonVec(p, "upper_tols", D, &zeros);
      t->lower_tols = jsonVec(p, "lower_tols", D, &zeros);
      t->first_step = jsonInt(p, "first_step", 0);
      t->last_step = jsonInt(p, "last_step", n_steps - 1);
      t->name = name;
      t->term_type = tt;
      return std::static_pointer_cast<TermInfo>(t);
    };
    if (typ == "joint_acc")
      return fill(std::make_shared<JointAccTermInfo>());
    return fill(std::make_shared<JointJerkTermInfo>());
  }
  if (typ == "joint_pos")
  {
    ensureOnlyMembers(p, { "coeffs", "first_step", "last_step", "targets", "lower_tols", "upper_tols", "use_time" }, typ);
    auto t = std::make_shared<JointPosTermInfo>();
    t->coeffs = jsonVec(p, "coeffs", D, &ones);
    t->targets = jsonVec(p, "targets", D, nullptr);
    t->upper_tols = jsonVec(p, "upper_tols", D, &zeros);
    t->lower_tols = jsonVec(p, "lower_tols", D, &zeros);
    t->first_step = jsonInt(p, "first_step", 0);
    t->last_step = jsonInt(p, "last_step", n_steps - 1);
    t->name = name;
    t->term_type = tt;
    return t;
  }
  if (typ == "cart_pose")
  {
    auto t = std::make_shared<CartPoseTermInfo>();
    t->source_frame = p["source_frame"].asString();
    t->target_frame = p["target_frame"].asString();
    if (t->source_frame != pci.kin->tip_link)
      printAndThrow("cart_pose source_frame " + t->source_frame + ": only the manipulator tip link " + pci.kin->tip_link + " is lowered");
    if (!pci.env->link_frames.count(t->target_frame))
      printAndThrow("cart_pose target_frame " + t->target_frame +
                    ": only static frames of the environment are lowered (DynamicCartPose is the reference's term for moving targets)");
    t->source_frame_offset = jsonOffset(p, "source_frame_offset_xyz", "source_frame_offset_wxyz");
    t->target_frame_offset = jsonOffset(p, "target_frame_offset_xyz", "target_frame_offset_wxyz");
    const DblVec one3(3, 1.0);
    const DblVec pc = jsonVec(p, "pos_coeffs", 3, &one3), rc = jsonVec(p, "rot_coeffs", 3, &one3);
    t->pos_coeffs = { { pc[0], pc[1], pc[2] } };
    t->rot_coeffs = { { rc[0], rc[1], rc[2] } };
    t->timestep = jsonInt(p, "timestep", n_steps - 1);
    t->name = name;
    t->term_type = tt;
    return t;
  }
  if (typ == "dynamic_cart_pose")
  {
    // DynamicCartPoseTermInfo::fromJson (problem_description.cpp:685-750): both frames are active links of the manipulator
    ensureOnlyMembers(p, { "timestep", "pos_coeffs", "rot_coeffs", "source_frame", "target_frame", "source_frame_offset_xyz",
                           "source_frame_offset_wxyz", "target_frame_offset_xyz", "target_frame_offset_wxyz" }, typ);
    auto t = std::make_shared<DynamicCartPoseTermInfo>();
    t->source_frame = p["source_frame"].asString();
    t->target_frame = p["target_frame"].asString();
    if (t->source_frame != pci.kin->tip_link)
      printAndThrow("dynamic_cart_pose source_frame " + t->source_frame + ": only the manipulator tip link " + pci.kin->tip_link + " is lowered");
    if (pci.kin->linkIndex(t->target_frame) < 0)
    {
      if (pci.env->link_frames.count(t->target_frame))
        printAndThrow("source '" + t->source_frame + "' and target '" + t->target_frame + "' are not both active links");  // :733-737
      printAndThrow("invalid target frame: " + t->target_frame);                                                          // :726-729
    }
    t->source_frame_offset = jsonOffset(p, "source_frame_offset_xyz", "source_frame_offset_wxyz");
    t->target_frame_offset = jsonOffset(p, "target_frame_offset_xyz", "target_frame_offset_wxyz");
    const DblVec one3(3, 1.0);
    const DblVec pc = jsonVec(p, "pos_coeffs", 3, &one3), rc = jsonVec(p, "rot_coeffs", 3, &one3);
    t->pos_coeffs = { { pc[0], pc[1], pc[2] } };
    t->rot_coeffs = { { rc[0], rc[1], rc[2] } };
    t->timestep = jsonInt(p, "timestep", n_steps - 1);
    t->name = name;
    t->term_type = tt;
    return t;
  }
  if (typ == "cart_vel")
  {
    // CartVelTermInfo::fromJson (problem_description.cpp:989-1009): all four fields are required
    for (const char* f : { "first_step", "last_step", "max_displacement", "link" })
      if (!p.isMember(f))
        printAndThrow(std::string("cart_vel: missing required field ") + f);
    ensureOnlyMembers(p, { "first_step", "last_step", "max_displacement", "link" }, typ);
    auto t = std::make_shared<CartVelTermInfo>();
    t->first_step = jsonInt(p, "first_step", 0);
    t->last_step = jsonInt(p, "last_step", 0);
    t->max_displacement = jsonDouble(p, "max_displacement", 0.0);
    t->link = p["link"].asString();
    t->name = name;
    t->term_type = tt;
    return t;
  }
  if (typ == "collision")
  {
    const int ev = jsonInt(p, "evaluator_type", 1);
    if (ev < 1 || ev > 4)
      printAndThrow("collision evaluator_type " + std::to_string(ev) + ": must be 1 .. 4");  // FAIL_IF_FALSE(<= 4), :1637; 0 = NONE
    const double lvs = jsonDouble(p, "longest_valid_segment_length", 0.5);
    if (!(lvs >= 0))
      printAndThrow("collision: longest_valid_segment_length must be >= 0");  // :1634
    if (p.isMember("pairs"))
      printAndThrow("collision per-pair margin overrides are not lowered by the device path");
    auto t = std::make_shared<CollisionTermInfo>();
    t->first_step = jsonInt(p, "first_step", 0);
    t->last_step = jsonInt(p, "last_step", n_steps - 1);
    if (!(0 <= t->first_step && t->first_step < n_steps && t->first_step <= t->last_step && t->last_step < n_steps))
      printAndThrow("collision: invalid first_step / last_step");  // FAIL_IF_FALSE, :1633-1634
    if (p.isMember("fixed_steps"))
      for (const auto& e : p["fixed_steps"].arr)
        t->fixed_steps.push_back(e.asInt());
    for (int fs : t->fixed_steps)
      if (fs < t->first_step || fs > t->last_step)
        printAndThrow("Fixed step " + std::to_string(fs) + " is not between first step " + std::to_string(t->first_step) +
                      " and last step " + std::to_string(t->last_step));
    const double buf = jsonDouble(p, "safety_margin_buffer", 0.5);  // quirk Q3: the JSON default is 0.5
    if (buf < 0)
      printAndThrow("collision: negative safety_margin_buffer");
    // ... and the reference's list of allowed fields (:1701-1711) does not contain "safety_margin_buffer": a JSON file that
    // supplies the key is rejected, the effective JSON-path buffer is always the default
    ensureOnlyMembers(p, { "type", "first_step", "last_step", "evaluator_type", "fixed_steps", "contact_test_type",
                           "longest_valid_segment_length", "coeffs", "dist_pen", "pairs" }, typ);
    t->config = TrajOptCollisionConfig(p["dist_pen"].asDouble(), p["coeffs"].asDouble());
    t->config.collision_margin_buffer = buf;
    t->config.type = static_cast<TrajOptCollisionConfig::CollisionEvaluatorType>(ev);
    t->config.longest_valid_segment_length = lvs;
    t->config.max_substates = 0;  // resolved from the initial trajectory by ProblemConstructionInfoFromJson
    t->name = name;
    t->term_type = tt;
    return t;
  }
  printAndThrow("term type \"" + typ + "\" is not lowered by the device path");
}
}  // namespace detail

/** ProblemConstructionInfo::fromJson (problem_description.cpp:269-308) for the lowered term classes */
inline ProblemConstructionInfo ProblemConstructionInfoFromJson(const json::Value& v, const std::shared_ptr<const Environment>& env)
{
  ProblemConstructionInfo pci(env);
  if (!v.isMember("basic_info"))
    printAndThrow("Json missing required section basic_info!");  // :280
  const json::Value& bi = v["basic_info"];
  pci.basic_info.n_steps = bi["n_steps"].asInt();
  pci.basic_info.manip = bi["manip"].asString();
  pci.resolveKin();  // "Manipulator does not exist: ..." (:292)
  pci.basic_info.use_time = bi.isMember("use_time") && bi["use_time"].asBool();
  pci.basic_info.dt_lower_lim = detail::jsonDouble(bi, "dt_lower_lim", 1.0);
  pci.basic_info.dt_upper_lim = detail::jsonDouble(bi, "dt_upper_lim", 1.0);
  if (pci.basic_info.dt_lower_lim <= 0 || pci.basic_info.dt_upper_lim < pci.basic_info.dt_lower_lim)
    printAndThrow("dt limits (Basic Info) invalid. The lower limit must be positive, and the minimum upper limit is equal to the lower limit.");
  const std::string solver = bi.isMember("convex_solver") ? bi["convex_solver"].asString() : "AUTO_SOLVER";
  if (solver != "AUTO_SOLVER" && solver != "OSQP")
    printAndThrow("convex_solver " + solver + ": the device QP solver restates the OSQP back-end only");
  if (bi.isMember("fixed_timesteps"))
    for (const auto& e : bi["fixed_timesteps"].arr)
      pci.basic_info.fixed_timesteps.push_back(e.asInt());
  if (bi.isMember("fixed_dofs"))
    for (const auto& e : bi["fixed_dofs"].arr)
      pci.basic_info.fixed_dofs.push_back(e.asInt());
  // readOptInfo (:147-166): known keys override the defaults, unknown keys are ignored
  if (v.isMember("opt_info"))
  {
    sco::BasicTrustRegionSQPParameters& o = pci.opt_info;
    for (const auto& kv : v["opt_info"].obj)
    {
      const std::string& k = kv.first;
      const json::Value& val = kv.second;
      if (k == "improve_ratio_threshold")
        o.improve_ratio_threshold = val.asDouble();
      else if (k == "min_trust_box_size")
        o.min_trust_box_size = val.asDouble();
      else if (k == "min_approx_improve")
        o.min_approx_improve = val.asDouble();
      else if (k == "min_approx_improve_frac")
        o.min_approx_improve_frac = val.asDouble();
      else if (k == "max_iter")
        o.max_iter = val.asDouble();
      else if (k == "trust_shrink_ratio")
        o.trust_shrink_ratio = val.asDouble();
      else if (k == "trust_expand_ratio")
        o.trust_expand_ratio = val.asDouble();
      else if (k == "cnt_tolerance")
        o.cnt_tolerance = val.asDouble();
      else if (k == "max_merit_coeff_increases")
        o.max_merit_coeff_increases = val.asDouble();
      else if (k == "merit_coeff_increase_ratio")
        o.merit_coeff_increase_ratio = val.asDouble();
      else if (k == "initial_merit_error_coeff")
        o.initial_merit_error_coeff = val.asDouble();
      else if (k == "inflate_constraints_individually")
        o.inflate_constraints_individually = (val.kind == json::Value::BOOL) ? val.asBool() : (val.asDouble() != 0.0);
      else if (k == "trust_box_size")
        o.trust_box_size = val.asDouble();
      else if (k == "max_time")
        o.max_time = val.asDouble();
    }
  }
  if (v.isMember("costs"))
    for (const auto& it : v["costs"].arr)
      pci.cost_infos.push_back(detail::readTerm(it, true, pci));
  if (v.isMember("constraints"))
    for (const auto& it : v["constraints"].arr)
      pci.cnt_infos.push_back(detail::readTerm(it, false, pci));
  for (const auto& lst : { pci.cost_infos, pci.cnt_infos })  // (:178-181, :208-211)
    for (const TermInfo::Ptr& ti : lst)
      if (static_cast<bool>(ti->term_type & TermType::TT_USE_TIME))
        pci.basic_info.use_time = true;
  // readInitInfo (:208-268)
  if (!v.isMember("init_info"))
    printAndThrow("Json missing required section init_info!");  // :306
  const json::Value& ii = v["init_info"];
  pci.init_info.dt = detail::jsonDouble(ii, "dt", 1.0);  // :226
  std::string typ = ii["type"].asString();
  std::transform(typ.begin(), typ.end(), typ.begin(), [](unsigned char c) { return static_cast<char>(std::tolower(c)); });
  const int D = static_cast<int>(pci.kin->numJoints());
  if (typ == "stationary")
    pci.init_info.type = InitInfo::STATIONARY;
  else if (typ == "given_traj")
  {
    pci.init_info.type = InitInfo::GIVEN_TRAJ;
    const auto& rows = ii["data"].arr;
    if (static_cast<int>(rows.size()) != pci.basic_info.n_steps)
      printAndThrow("given initialization traj has wrong length");  // :241
    pci.init_info.data = TrajArray(pci.basic_info.n_steps, D);
    for (int t = 0; t < pci.basic_info.n_steps; ++t)
    {
      if (static_cast<int>(rows[static_cast<std::size_t>(t)].arr.size()) != D)
        printAndThrow("given initialization traj has wrong number of dof values");
      for (int j = 0; j < D; ++j)
        pci.init_info.data(t, j) = rows[static_cast<std::size_t>(t)].arr[static_cast<std::size_t>(j)].asDouble();
    }
  }
  else if (typ == "joint_interpolated")
  {
    pci.init_info.type = InitInfo::JOINT_INTERPOLATED;
    const auto& e = ii["endpoint"].arr;
    if (static_cast<int>(e.size()) != D)
      printAndThrow("wrong number of dof values in initialization. expected " + std::to_string(D) + " got " + std::to_string(e.size()));  // :259
    pci.init_info.data = TrajArray(1, D);
    for (int j = 0; j < D; ++j)
      pci.init_info.data(0, j) = e[static_cast<std::size_t>(j)].asDouble();
  }
  else
    printAndThrow("init_info did not have a valid type from Json. Valid types are stationary, joint_interpolated, or given_traj");  // :267
  return pci;
}

/** ConstructProblem(const Json::Value&, env) (problem_description.cpp:532-550) */
inline TrajOptProb::Ptr ConstructProblem(const std::string& json_text, const std::shared_ptr<const Environment>& env,
                                         sco::BasicTrustRegionSQPParameters* opt_info_out = nullptr)
{
  const ProblemConstructionInfo pci = ProblemConstructionInfoFromJson(json::parse(json_text), env);
  if (opt_info_out)
    *opt_info_out = pci.opt_info;
  return ConstructProblem(pci);
}
}  // namespace trajopt
}  // namespace tmx

#endif  // TMX_TRAJOPT_JSON_HPP_
