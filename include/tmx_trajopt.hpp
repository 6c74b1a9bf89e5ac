// tmx_trajopt.hpp — C++ host side above the C-ABI of libtrajopt_mi355x.so (include/tmx.h).
//
// Header-only mirror of the reference's caller-facing interface for the SQP hot path, so that a trajopt user finds
// the same names, argument meaning and error behaviour (std::runtime_error where the reference PRINT_AND_THROWs):
//
//   tmx::trajopt::  TermType, BasicInfo, InitInfo, TermInfo, JointPosTermInfo, JointVelTermInfo, JointAccTermInfo, JointJerkTermInfo, CartPoseTermInfo, DynamicCartPoseTermInfo, AvoidSingularityTermInfo,
//                   CollisionTermInfo, ProblemConstructionInfo, TrajOptProb, ConstructProblem
//                     <- trajopt/include/trajopt/problem_description.hpp:29-66, 68-107, 123-160, 162-186, 199-230,
//                        235-262, 352-392, 421-514, 597-617, 661-663 ; trajopt/src/problem_description.cpp:410-592
//   tmx::sco::      OptStatus, BasicTrustRegionSQPParameters, OptResults, BasicTrustRegionSQPBatchedHip
//                     <- trajopt_sco/include/trajopt_sco/optimizers.hpp:25-33, 40-59, 92-135, 137-218 ;
//                        trajopt_sco/src/optimizers.cpp:120-136, 699-991
//
// What differs from the reference, on purpose:
//   * tesseract is not a dependency: `JointGroup` / `Environment` below are plain-data stand-ins for the few things
//     ConstructProblem and the TermInfo::hatch functions read from tesseract (chain description, joint limits, current
//     state, static link frames, sphere collision geometry).  Inside the trajopt tree the adapter of INTEGRATION.md §1
//     fills them from the real tesseract objects.
//   * matrices are plain row-major arrays (`TrajArray`), no Eigen.
//   * hatch() lowers a term to the flat `tmx_term` of the C-ABI instead of creating sco::Cost / sco::Constraint objects;
//     a term (or option) the device path does not lower throws — there is NO CPU fallback behind this interface.
//   * the optimizer runs a BATCH of seeds (one trajectory problem per workgroup); the single-seed calls of the reference
//     (`initialize(DblVec)`, `x()`, `results()`) are the batch-of-one case and report the best seed otherwise.
//   * JSON (`ProblemConstructionInfo::fromJson`) is handled by trajopt_amd/json_io.py in this repository (no JSON library
//     in the C++ toolchain here); inside the trajopt tree the reference's own fromJson fills these structs.
// This file contains no arithmetic of the hot path: everything numerical happens behind tmx_sqp_run().
#ifndef TMX_TRAJOPT_HPP_
#define TMX_TRAJOPT_HPP_

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "tmx.h"

namespace tmx
{
using DblVec = std::vector<double>;
using IntVec = std::vector<int>;

[[noreturn]] inline void printAndThrow(const std::string& msg) { throw std::runtime_error(msg); }  // macros.h:90-98

/** row-major dense matrix: trajopt::TrajArray (trajopt/include/trajopt/typedefs.hpp) without Eigen */
struct TrajArray
{
  int rows_{ 0 }, cols_{ 0 };
  DblVec data;
  TrajArray() = default;
  TrajArray(int r, int c, double v = 0.0) : rows_(r), cols_(c), data(static_cast<std::size_t>(r) * static_cast<std::size_t>(c), v) {}
  int rows() const { return rows_; }
  int cols() const { return cols_; }
  double& operator()(int r, int c) { return data[static_cast<std::size_t>(r) * static_cast<std::size_t>(cols_) + static_cast<std::size_t>(c)]; }
  double operator()(int r, int c) const { return data[static_cast<std::size_t>(r) * static_cast<std::size_t>(cols_) + static_cast<std::size_t>(c)]; }
};

/** rigid transform, row-major 3x4 [R | t] (the layout of tmx_joint::origin / tmx_term::target_pose) */
struct Transform
{
  std::array<double, 12> m{ { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0 } };
  static Transform Identity() { return Transform(); }
  static Transform Translation(double x, double y, double z)
  {
    Transform t;
    t.m[3] = x;
    t.m[7] = y;
    t.m[11] = z;
    return t;
  }
  /** Eigen::Quaterniond(w,x,y,z) (normalised) + translation */
  static Transform FromQuaternion(double w, double x, double y, double z, double tx = 0, double ty = 0, double tz = 0)
  {
    const double n = std::sqrt(w * w + x * x + y * y + z * z);
    if (n == 0.0)
      printAndThrow("zero quaternion");
    w /= n, x /= n, y /= n, z /= n;
    Transform t;
    t.m = { { 1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), tx,  //
              2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w), ty,  //
              2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y), tz } };
    return t;
  }
  Transform operator*(const Transform& b) const
  {
    Transform o;
    for (int r = 0; r < 3; ++r)
    {
      for (int c = 0; c < 3; ++c)
        o.m[r * 4 + c] = m[r * 4 + 0] * b.m[0 + c] + m[r * 4 + 1] * b.m[4 + c] + m[r * 4 + 2] * b.m[8 + c];
      o.m[r * 4 + 3] = m[r * 4 + 0] * b.m[3] + m[r * 4 + 1] * b.m[7] + m[r * 4 + 2] * b.m[11] + m[r * 4 + 3];
    }
    return o;
  }
  bool isIdentity(double eps = 1e-12) const
  {
    const Transform i;
    for (std::size_t k = 0; k < 12; ++k)
      if (std::fabs(m[k] - i.m[k]) > eps)
        return false;
    return true;
  }
};

namespace trajopt
{
/** problem_description.hpp:29-66 */
enum class TermType : char
{
  TT_INVALID = 0,
  TT_COST = 0x1,
  TT_CNT = 0x2,
  TT_USE_TIME = 0x4
};
inline TermType operator|(TermType a, TermType b) { return static_cast<TermType>(static_cast<char>(a) | static_cast<char>(b)); }
inline TermType operator&(TermType a, TermType b) { return static_cast<TermType>(static_cast<char>(a) & static_cast<char>(b)); }
inline TermType operator~(TermType a) { return static_cast<TermType>(~static_cast<char>(a)); }

/** stand-in for tesseract::kinematics::JointGroup: a serial chain with sphere collision geometry */
struct JointGroup
{
  using ConstPtr = std::shared_ptr<const JointGroup>;
  std::vector<std::string> joint_names;
  std::vector<tmx_joint> joints;  // type, fixed parent->joint origin, axis
  DblVec lower, upper;            // getLimits().joint_limits
  Transform base;                 // world_T_base
  Transform tool;                 // last link -> tip_link frame (tcp)
  std::string tip_link;           // the link the tool frame is attached to
  std::vector<std::string> link_names;  // optional: child link of every joint (DynamicCartPose / AvoidSingularity name moving links)
  /** index of a moving link (child of joint k), -1 when the name is not one */
  int linkIndex(const std::string& link) const
  {
    for (std::size_t k = 0; k < link_names.size() && k < joints.size(); ++k)
      if (link_names[k] == link)
        return static_cast<int>(k);
    return link == tip_link ? static_cast<int>(joints.size()) - 1 : -1;
  }
  std::vector<tmx_link_sphere> link_spheres;
  std::vector<double> link_sphere_axes;  // ... or capsules: 3 per link sphere, link frame (swept from centre to centre + axis); empty = all spheres
  // ... or convex hulls (round 4): 2 per link primitive (first vertex, number of vertices; 0 = sphere / capsule) into hull_vertices,
  // 3 doubles per vertex in the link frame; the primitive's radius rounds the hull (tmx_problem_desc::link_hull; GJK / EPA contacts)
  std::vector<int32_t> link_hull;
  std::vector<double> hull_vertices;
  std::size_t numJoints() const { return joints.size(); }
};

/** stand-in for the parts of tesseract::environment::Environment the problem construction consults */
struct Environment
{
  using ConstPtr = std::shared_ptr<const Environment>;
  std::map<std::string, std::shared_ptr<const JointGroup>> manipulators;  // getJointGroup(manip)
  std::map<std::string, DblVec> state;                                    // getState(): manip -> current joint values
  std::map<std::string, Transform> link_frames;                           // static links: world_T_link
  std::vector<tmx_obstacle_sphere> obstacles;                             // world collision geometry: spheres ...
  std::vector<double> obstacle_axes;  // ... or capsules: 3 per obstacle (swept from centre to centre + axis); empty = all spheres
  std::vector<double> obstacle_boxes;  // ... or rounded boxes: 12 per obstacle (half extents, rotation world_R_box row-major); empty = none
  std::vector<int32_t> obstacle_mesh;  // ... or convex triangle meshes: 2 per obstacle (first triangle, number of triangles); empty = none
  std::vector<double> mesh_triangles;  // 9 per triangle: three world-frame vertices, counter-clockwise seen from outside
  std::shared_ptr<const JointGroup> getJointGroup(const std::string& manip) const
  {
    auto it = manipulators.find(manip);
    if (it == manipulators.end())
      printAndThrow("Manipulator does not exist: " + manip);  // problem_description.cpp:292
    return it->second;
  }
};

/** problem_description.hpp:123-160 */
struct BasicInfo
{
  int n_steps{ -1 };
  std::string manip;
  IntVec fixed_timesteps;
  IntVec fixed_dofs;
  bool use_time{ false };  // one (1/dt) variable per step behind the joints (problem_description.hpp:150); dense QP engine when a TotalTime term or a squared velocity cost uses it
  double dt_upper_lim{ 1.0 };
  double dt_lower_lim{ 1.0 };
};

/** problem_description.hpp:162-186 */
struct InitInfo
{
  enum Type : std::uint8_t
  {
    STATIONARY,
    JOINT_INTERPOLATED,
    GIVEN_TRAJ
  };
  Type type{ STATIONARY };
  TrajArray data;
  double dt{ 1.0 };
};

class TrajOptProb;

/** problem_description.hpp:199-230 */
struct TermInfo
{
  using Ptr = std::shared_ptr<TermInfo>;
  std::string name;
  TermType term_type{ TermType::TT_INVALID };
  explicit TermInfo(TermType supported) : supported_term_types_(supported) {}
  virtual ~TermInfo() = default;
  TermType getSupportedTypes() const { return supported_term_types_; }
  /** lowers the term into the problem's flat term table (the reference creates sco::Cost / sco::Constraint objects) */
  virtual void hatch(TrajOptProb& prob) = 0;

private:
  TermType supported_term_types_;
};

}  // namespace trajopt

namespace sco
{
/** optimizers.hpp:25-33 — same integer values as tmx_opt_status */
enum class OptStatus : std::uint8_t
{
  OPT_CONVERGED = TMX_OPT_CONVERGED,
  OPT_SCO_ITERATION_LIMIT = TMX_OPT_SCO_ITERATION_LIMIT,
  OPT_PENALTY_ITERATION_LIMIT = TMX_OPT_PENALTY_ITERATION_LIMIT,
  OPT_TIME_LIMIT = TMX_OPT_TIME_LIMIT,
  OPT_FAILED = TMX_OPT_FAILED,
  INVALID = TMX_OPT_INVALID
};
inline const char* statusToString(OptStatus s)  // optimizers.cpp:33-50
{
  static const char* const names[] = { "CONVERGED", "SCO_ITERATION_LIMIT", "PENALTY_ITERATION_LIMIT", "TIME_LIMIT", "FAILED", "INVALID" };
  return names[std::min<std::size_t>(static_cast<std::size_t>(s), 5)];
}

/** optimizers.hpp:92-135 (same defaults).  log_results / log_dir / num_threads have no meaning for a batched launch and are
    not mirrored; max_time is the wall-clock limit of optimize() in seconds (optimizers.cpp:738-753), tested on the device
    at the top of every SQP iteration. */
struct BasicTrustRegionSQPParameters
{
  double improve_ratio_threshold{ 0.25 };
  double min_trust_box_size{ 1e-4 };
  double min_approx_improve{ 1e-4 };
  double min_approx_improve_frac{ std::numeric_limits<double>::lowest() };
  double max_iter{ 50 };
  double trust_shrink_ratio{ 0.1 };
  double trust_expand_ratio{ 1.5 };
  double cnt_tolerance{ 1e-4 };
  double max_merit_coeff_increases{ 5 };
  int max_qp_solver_failures{ 3 };
  double merit_coeff_increase_ratio{ 10 };
  double initial_merit_error_coeff{ 10 };
  bool inflate_constraints_individually{ true };
  double trust_box_size{ 1e-1 };
  double max_time{ std::numeric_limits<double>::max() };

  tmx_sqp_params toTmx() const
  {
    tmx_sqp_params p{};
    p.improve_ratio_threshold = improve_ratio_threshold;
    p.min_trust_box_size = min_trust_box_size;
    p.min_approx_improve = min_approx_improve;
    p.min_approx_improve_frac = min_approx_improve_frac;
    p.max_iter = static_cast<int32_t>(max_iter);
    p.max_qp_solver_failures = max_qp_solver_failures;
    p.trust_shrink_ratio = trust_shrink_ratio;
    p.trust_expand_ratio = trust_expand_ratio;
    p.cnt_tolerance = cnt_tolerance;
    p.max_merit_coeff_increases = max_merit_coeff_increases;
    p.merit_coeff_increase_ratio = merit_coeff_increase_ratio;
    p.initial_merit_error_coeff = initial_merit_error_coeff;
    p.inflate_constraints_individually = inflate_constraints_individually ? 1 : 0;
    p.trust_box_size = trust_box_size;
    p.max_time = max_time;
    return p;
  }
};

/** optimizers.hpp:40-59 */
struct OptResults
{
  DblVec x;
  OptStatus status{ OptStatus::INVALID };
  double total_cost{ 0 };
  DblVec cost_vals;
  DblVec cnt_viols;
  int n_func_evals{ 0 };
  int n_qp_solves{ 0 };
  void clear()
  {
    x.clear();
    status = OptStatus::INVALID;
    cost_vals.clear();
    cnt_viols.clear();
    n_func_evals = 0;
    n_qp_solves = 0;
    total_cost = 0;
  }
};
}  // namespace sco

namespace trajopt
{
/** problem_description.hpp:235-262 */
struct ProblemConstructionInfo
{
  BasicInfo basic_info;
  sco::BasicTrustRegionSQPParameters opt_info;  // quirk Q6: unused by the reference's own ConstructProblem, same here
  std::vector<TermInfo::Ptr> cost_infos;
  std::vector<TermInfo::Ptr> cnt_infos;
  InitInfo init_info;
  std::shared_ptr<const Environment> env;
  std::shared_ptr<const JointGroup> kin;

  explicit ProblemConstructionInfo(std::shared_ptr<const Environment> env_) : env(std::move(env_)) {}
  /** readBasicInfo's tail (problem_description.cpp:288-293): resolve `kin` from basic_info.manip */
  void resolveKin() { kin = env->getJointGroup(basic_info.manip); }
};

/** problem_description.hpp:68-107: the hatched problem.  Holds the flat description the C-ABI consumes. */
class TrajOptProb
{
public:
  using Ptr = std::shared_ptr<TrajOptProb>;
  TrajOptProb(int n_steps, const ProblemConstructionInfo& pci) : n_steps_(n_steps), kin_(pci.kin), env_(pci.env)
  {
    if (!kin_)
      printAndThrow("ProblemConstructionInfo.kin is not set (call resolveKin())");
    if (kin_->numJoints() > TMX_MAX_DOF)
      printAndThrow("n_dof exceeds TMX_MAX_DOF");
  }
  int GetNumSteps() const { return n_steps_; }
  int GetNumDOF() const { return static_cast<int>(kin_->numJoints()); }
  bool GetHasTime() const { return has_time_; }
  void SetHasTime(bool tmp) { has_time_ = tmp; }
  void setTimeLimits(double lower, double upper)
  {
    dt_lower_lim_ = lower;
    dt_upper_lim_ = upper;
  }
  /** number of variables per step: the joints, plus the time column of a time-parameterised problem (TrajOptProb::GetNumDOF of the
      reference counts the same, problem_description.hpp:300) */
  int GetNumVarsPerStep() const { return GetNumDOF() + (has_time_ ? 1 : 0); }
  std::shared_ptr<const JointGroup> GetKin() const { return kin_; }
  std::shared_ptr<const Environment> GetEnv() const { return env_; }
  const TrajArray& GetInitTraj() const { return init_traj_; }
  void SetInitTraj(const TrajArray& t) { init_traj_ = t; }
  std::size_t getNumCosts() const { return n_cost_terms_; }
  std::size_t getNumConstraints() const { return terms_.size() - n_cost_terms_; }

  /** names of the costs / constraints the terms expand to, in the order of OptResults::cost_vals / cnt_viols: costs in
      hatch order; constraints with all equalities in front of the inequalities (sco::OptProb, modeling.cpp:234-241) */
  const std::vector<std::string>& getCostNames() const { return cost_names_; }
  std::vector<std::string> getCntNames() const
  {
    std::vector<std::string> out(eq_cnt_names_);
    out.insert(out.end(), ineq_cnt_names_.begin(), ineq_cnt_names_.end());
    return out;
  }

  /** used by TermInfo::hatch: costs are hatched before constraints (problem_description.cpp:560-571).  `name` is the
      TermInfo name; collision terms expand to one cost / constraint "name_<step>" per non-fixed step (:1773, :1833) */
  /** a function term: the program is copied into the problem (the descriptor points at the copy) */
  void addFuncTerm(tmx_term t, const std::vector<int32_t>& ops, const DblVec& consts, int n_outputs, const std::string& name,
                   const IntVec& fixed_steps = {})
  {
    auto prog = std::make_shared<FuncProgram>();
    prog->ops = ops;
    prog->consts = consts;
    prog->e.n_ops = static_cast<int32_t>(ops.size() / 2);
    prog->e.n_consts = static_cast<int32_t>(consts.size());
    prog->e.n_outputs = n_outputs;
    prog->e.ops = prog->ops.data();
    prog->e.consts = prog->consts.data();
    programs_.push_back(prog);
    t.expr = &prog->e;
    addTerm(t, fixed_steps, name);
  }
  void addTerm(const tmx_term& t, const IntVec& fixed_steps = {}, const std::string& name = std::string())
  {
    std::vector<std::string> names;
    if (t.kind == TMX_TERM_COLLISION_COST || t.kind == TMX_TERM_COLLISION_CNT)
    {
      if (t.evaluator_type >= 2)  // one term per segment (problem_description.cpp:1723, :1781)
        for (int i = t.first_step; i < t.last_step; ++i)
          names.push_back(name + "_" + std::to_string(i));
      else
        for (int i = t.first_step; i <= t.last_step; ++i)
          if (std::find(fixed_steps.begin(), fixed_steps.end(), i) == fixed_steps.end())
            names.push_back(name + "_" + std::to_string(i));
    }
    else if (t.kind == TMX_TERM_CART_VEL)  // one cost named after the term / one constraint "CartVel" per step (:1029-1050)
      for (int i = t.first_step; i <= t.last_step; ++i)
        names.push_back(t.is_constraint ? std::string("CartVel") : name);
    else if (t.kind == TMX_TERM_JOINT_VEL_TIME)  // one cost / constraint per joint, name_j<j> (problem_description.cpp:1267-1283)
      for (int j = 0; j < GetNumDOF(); ++j)
        names.push_back(name + "_j" + std::to_string(j));
    else if (t.kind == TMX_TERM_AVOID_SINGULARITY)  // name_<step> (problem_description.cpp:1924)
      for (int i = t.first_step; i <= t.last_step; ++i)
        names.push_back(name + "_" + std::to_string(i));
    else if (t.kind == TMX_TERM_FUNC_COST || t.kind == TMX_TERM_FUNC_CNT || t.kind == TMX_TERM_FUNC_ERR_COST)
    {
      // one sco cost / constraint per step; UserDefinedTermInfo appends the step (problem_description.cpp:611-630)
      for (int i = t.first_step; i <= t.last_step; ++i)
        if (std::find(fixed_steps.begin(), fixed_steps.end(), i) == fixed_steps.end())
          names.push_back(t.kind == TMX_TERM_FUNC_COST ? name : name + "_" + std::to_string(i));
    }
    else
      names.push_back(name);
    const bool ineq = t.kind == TMX_TERM_JOINT_POS_INEQ_CNT || t.kind == TMX_TERM_COLLISION_CNT || t.kind == TMX_TERM_JOINT_VEL_INEQ_CNT ||
                      t.kind == TMX_TERM_JOINT_ACC_INEQ_CNT || t.kind == TMX_TERM_JOINT_JERK_INEQ_CNT || t.kind == TMX_TERM_CART_VEL ||
                      (t.kind == TMX_TERM_FUNC_CNT && t.cnt_type == 1) || t.kind == TMX_TERM_AVOID_SINGULARITY ||
                      (t.kind == TMX_TERM_TOTAL_TIME && !(std::fabs(t.margin) < 1e-5)) || (t.kind == TMX_TERM_JOINT_VEL_TIME && time_ineq_);
    time_ineq_ = false;
    std::vector<std::string>& dst = !t.is_constraint ? cost_names_ : (ineq ? ineq_cnt_names_ : eq_cnt_names_);
    dst.insert(dst.end(), names.begin(), names.end());
    terms_.push_back(t);
    term_fixed_.push_back(fixed_steps);
    if (!t.is_constraint)
    {
      if (n_cost_terms_ + 1 != terms_.size())
        printAndThrow("costs must be hatched before constraints");
      ++n_cost_terms_;
    }
  }
  /** the next TMX_TERM_JOINT_VEL_TIME constraint handed to addTerm has non-zero tolerances (an inequality: its names go behind the equalities) */
  void nextTimeTermIsIneq() { time_ineq_ = true; }
  void setFixed(const IntVec& steps, const IntVec& dofs)
  {
    fixed_steps_.assign(steps.begin(), steps.end());
    fixed_dofs_.assign(dofs.begin(), dofs.end());
  }

  /** the flat description; pointers stay valid as long as this object lives and is not modified */
  const tmx_problem_desc& desc()
  {
    tmx_problem_desc d{};
    const int D = GetNumDOF();
    d.n_dof = D;
    d.n_steps = n_steps_;
    for (int j = 0; j < D; ++j)
    {
      d.joint_lower[j] = kin_->lower[static_cast<std::size_t>(j)];
      d.joint_upper[j] = kin_->upper[static_cast<std::size_t>(j)];
      d.joints[j] = kin_->joints[static_cast<std::size_t>(j)];
    }
    std::copy(kin_->base.m.begin(), kin_->base.m.end(), d.base);
    std::copy(kin_->tool.m.begin(), kin_->tool.m.end(), d.tool);
    d.n_link_spheres = static_cast<int32_t>(kin_->link_spheres.size());
    d.link_spheres = kin_->link_spheres.data();
    d.link_sphere_axes = (kin_->link_sphere_axes.size() == 3 * kin_->link_spheres.size() && !kin_->link_spheres.empty()) ? kin_->link_sphere_axes.data() : nullptr;
    if (kin_->link_hull.size() == 2 * kin_->link_spheres.size() && !kin_->hull_vertices.empty())
    {
      d.link_hull = kin_->link_hull.data();
      d.hull_vertices = kin_->hull_vertices.data();
      d.n_hull_vertices = static_cast<int32_t>(kin_->hull_vertices.size() / 3);
    }
    d.n_obstacles = static_cast<int32_t>(env_ ? env_->obstacles.size() : 0);
    d.obstacles = env_ ? env_->obstacles.data() : nullptr;
    d.obstacle_axes = (env_ && env_->obstacle_axes.size() == 3 * env_->obstacles.size() && !env_->obstacles.empty()) ? env_->obstacle_axes.data() : nullptr;
    d.obstacle_boxes = (env_ && env_->obstacle_boxes.size() == 12 * env_->obstacles.size() && !env_->obstacles.empty()) ? env_->obstacle_boxes.data() : nullptr;
    if (env_ && env_->obstacle_mesh.size() == 2 * env_->obstacles.size() && !env_->mesh_triangles.empty())
    {
      d.obstacle_mesh = env_->obstacle_mesh.data();
      d.mesh_triangles = env_->mesh_triangles.data();
      d.n_mesh_triangles = static_cast<int32_t>(env_->mesh_triangles.size() / 9);
    }
    d.n_fixed_steps = static_cast<int32_t>(fixed_steps_.size());
    d.fixed_steps = fixed_steps_.data();
    d.n_fixed_dofs = static_cast<int32_t>(fixed_dofs_.size());
    d.fixed_dofs = fixed_dofs_.data();
    for (std::size_t k = 0; k < terms_.size(); ++k)
    {
      term_fixed32_.resize(terms_.size());
      term_fixed32_[k].assign(term_fixed_[k].begin(), term_fixed_[k].end());
      terms_[k].n_fixed_steps = static_cast<int32_t>(term_fixed32_[k].size());
      terms_[k].fixed_steps = term_fixed32_[k].empty() ? nullptr : term_fixed32_[k].data();
    }
    d.n_terms = static_cast<int32_t>(terms_.size());
    d.terms = terms_.data();
    d.use_time = has_time_ ? 1 : 0;
    d.dt_lower_lim = dt_lower_lim_;
    d.dt_upper_lim = dt_upper_lim_;
    desc_ = d;
    return desc_;
  }

private:
  int n_steps_;
  std::shared_ptr<const JointGroup> kin_;
  std::shared_ptr<const Environment> env_;
  TrajArray init_traj_;
  std::vector<tmx_term> terms_;
  std::vector<IntVec> term_fixed_;
  std::vector<std::vector<int32_t>> term_fixed32_;
  std::size_t n_cost_terms_{ 0 };
  std::vector<int32_t> fixed_steps_, fixed_dofs_;
  bool has_time_{ false }, time_ineq_{ false };
  double dt_lower_lim_{ 1.0 }, dt_upper_lim_{ 1.0 };
  std::vector<std::string> cost_names_, eq_cnt_names_, ineq_cnt_names_;
  struct FuncProgram
  {
    std::vector<int32_t> ops;
    DblVec consts;
    tmx_expr e{};
  };
  std::vector<std::shared_ptr<FuncProgram>> programs_;
  tmx_problem_desc desc_{};
};

namespace detail
{
/** checkParameterSize (problem_description.cpp:70-89): a single value is broadcast, anything else must match */
inline void checkParameterSize(DblVec& parameter, std::size_t expected_size, const std::string& name, bool apply_first = true)
{
  if (apply_first && parameter.size() == 1)
    parameter = DblVec(expected_size, parameter[0]);
  else if (parameter.size() != expected_size)
    printAndThrow("wrong number of " + name + ". expected " + std::to_string(expected_size) + " got " + std::to_string(parameter.size()));
}
inline bool doubleEquals(double a, double b, double eps = 1e-5) { return std::fabs(a - b) < eps; }  // trajopt_common/utils.hpp
inline bool allZero(const DblVec& v)
{
  return std::all_of(v.begin(), v.end(), [](double x) { return doubleEquals(x, 0.); });
}
inline tmx_term blankTerm()
{
  tmx_term t{};
  return t;
}
/** validateTolerances (trajopt/src/kinematic_terms.cpp:41-55); six values each (the rows of calcTransformError) or none */
inline void validateTolerances(const std::string& who, const DblVec& lower, const DblVec& upper)
{
  if (lower.size() != upper.size())
    printAndThrow(who + ": Mismatched tolerance sizes. lower: " + std::to_string(lower.size()) + ", upper: " + std::to_string(upper.size()));
  if (!lower.empty() && lower.size() != 6)
    printAndThrow(who + ": pose tolerances have six values");
  for (std::size_t i = 0; i < lower.size(); ++i)
    if (lower[i] > upper[i])
      printAndThrow(who + ": Inverted tolerance band - lower > upper at one or more indices.");
}
inline void fillTolerances(tmx_term& t, const DblVec& lower, const DblVec& upper)
{
  for (std::size_t i = 0; i < lower.size() && i < 6; ++i)
  {
    t.lower_tols[i] = lower[i];
    t.upper_tols[i] = upper[i];
  }
}
}  // namespace detail

/** problem_description.hpp:421-469 ; hatch: problem_description.cpp:1073-1176 */
struct JointPosTermInfo : public TermInfo
{
  DblVec coeffs;
  DblVec targets;
  DblVec upper_tols;
  DblVec lower_tols;
  int first_step = 0;
  int last_step = -1;
  JointPosTermInfo() : TermInfo(TermType::TT_COST | TermType::TT_CNT | TermType::TT_USE_TIME) {}
  void hatch(TrajOptProb& prob) override
  {
    const std::size_t n_dof = prob.GetKin()->numJoints();
    if (coeffs.empty())
      coeffs = DblVec(n_dof, 1);
    if (upper_tols.empty())
      upper_tols = DblVec(n_dof, 0);
    if (lower_tols.empty())
      lower_tols = DblVec(n_dof, 0);
    if (last_step <= -1)
      last_step = prob.GetNumSteps() - 1;
    // step range handling as problem_description.cpp:1092-1108: clamp to the last step, reversed ranges are swapped
    if ((prob.GetNumSteps() - 1) <= first_step)
      first_step = prob.GetNumSteps() - 1;
    if ((prob.GetNumSteps() - 1) <= last_step)
      last_step = prob.GetNumSteps() - 1;
    if (last_step < first_step)
      std::swap(first_step, last_step);
    detail::checkParameterSize(coeffs, n_dof, "JointPosTermInfo coeffs");
    detail::checkParameterSize(targets, n_dof, "JointPosTermInfo targets");
    detail::checkParameterSize(upper_tols, n_dof, "JointPosTermInfo upper_tols");
    detail::checkParameterSize(lower_tols, n_dof, "JointPosTermInfo lower_tols");
    const bool zero = detail::allZero(upper_tols) && detail::allZero(lower_tols);
    tmx_term t = detail::blankTerm();
    t.first_step = first_step;
    t.last_step = last_step;
    for (std::size_t j = 0; j < n_dof; ++j)
    {
      t.coeffs[j] = coeffs[j];
      t.targets[j] = targets[j];
      t.upper_tols[j] = upper_tols[j];
      t.lower_tols[j] = lower_tols[j];
    }
    if (static_cast<bool>(term_type & TermType::TT_COST))
    {
      t.kind = zero ? TMX_TERM_JOINT_POS_EQ_COST : TMX_TERM_JOINT_POS_INEQ_COST;  // :1128-1149
      t.is_constraint = 0;
    }
    else if (static_cast<bool>(term_type & TermType::TT_CNT))
    {
      t.kind = zero ? TMX_TERM_JOINT_POS_EQ_CNT : TMX_TERM_JOINT_POS_INEQ_CNT;  // :1150-1171
      t.is_constraint = 1;
    }
    else
      return;  // "JointPosTermInfo does not have a valid term_type defined. No cost/constraint applied" (:1172-1175)
    prob.addTerm(t, {}, name);
  }
};

/** problem_description.hpp:471-514 ; hatch: problem_description.cpp:1197-1372 without time parameterisation: the squared
    cost (JointVelEqCost), the equality constraint and the hinge cost / inequality constraint forms. */
struct JointVelTermInfo : public TermInfo
{
  DblVec coeffs;
  DblVec targets;
  DblVec upper_tols;
  DblVec lower_tols;
  int first_step = 0;
  int last_step = -1;
  JointVelTermInfo() : TermInfo(TermType::TT_COST | TermType::TT_CNT | TermType::TT_USE_TIME) {}
  void hatch(TrajOptProb& prob) override
  {
    const std::size_t n_dof = prob.GetKin()->numJoints();
    if (coeffs.empty())
      coeffs = DblVec(n_dof, 1);
    if (upper_tols.empty())
      upper_tols = DblVec(n_dof, 0);
    if (lower_tols.empty())
      lower_tols = DblVec(n_dof, 0);
    if (last_step <= -1)
      last_step = prob.GetNumSteps() - 1;
    if ((prob.GetNumSteps() - 2) <= first_step)  // :1212-1226
      first_step = prob.GetNumSteps() - 2;
    if ((prob.GetNumSteps() - 1) <= last_step)
      last_step = prob.GetNumSteps() - 1;
    if (last_step == first_step)
      last_step += 1;
    if (last_step < first_step)
      std::swap(first_step, last_step);
    detail::checkParameterSize(coeffs, n_dof, "JointVelTermInfo coeffs");
    detail::checkParameterSize(targets, n_dof, "JointVelTermInfo targets");
    detail::checkParameterSize(upper_tols, n_dof, "JointVelTermInfo upper_tols");
    detail::checkParameterSize(lower_tols, n_dof, "JointVelTermInfo lower_tols");
    if (first_step < 0)
      printAndThrow("JointVelEqCost, trajectory is too short!");  // trajectory_costs.cpp:269-270
    const bool zero = detail::allZero(upper_tols) && detail::allZero(lower_tols);
    tmx_term t = detail::blankTerm();
    if (static_cast<bool>(term_type & TermType::TT_USE_TIME))
    {
      // :1244-1325: per joint one TrajOptCostFromErrFunc (SQUARED / HINGE) or TrajOptConstraintFromErrFunc (EQ / INEQ) over
      // JointVelErrCalculator / JointVelJacCalculator (kinematic_terms.cpp:427-470); lowered as one term, expanded per joint at upload
      if (!prob.GetHasTime())
        printAndThrow(name + " uses time but the problem has no time variables (basic_info.use_time)");
      t.kind = TMX_TERM_JOINT_VEL_TIME;
      t.is_constraint = static_cast<bool>(term_type & TermType::TT_COST) ? 0 : 1;
      t.first_step = first_step;
      t.last_step = last_step;
      for (std::size_t j = 0; j < n_dof; ++j)
      {
        t.coeffs[j] = coeffs[j];
        t.targets[j] = targets[j];
        t.upper_tols[j] = upper_tols[j];
        t.lower_tols[j] = lower_tols[j];
      }
      if (t.is_constraint && !zero)
        prob.nextTimeTermIsIneq();
      prob.addTerm(t, {}, name);
      return;
    }
    // :1246-1372 without use_time: zero tolerances -> JointVelEqCost / JointVelEqConstraint, else JointVelIneqCost /
    // JointVelIneqConstraint.  The constraint and hinge forms put rows on two consecutive waypoints; a library built without
    // TMX_LINK_ROWS refuses them at upload (TMX_ERR_UNSUPPORTED), which optimize() reports as an exception.
    if (static_cast<bool>(term_type & TermType::TT_COST))
    {
      t.kind = zero ? TMX_TERM_JOINT_VEL_COST : TMX_TERM_JOINT_VEL_INEQ_COST;
      t.is_constraint = 0;
    }
    else if (static_cast<bool>(term_type & TermType::TT_CNT))
    {
      t.kind = zero ? TMX_TERM_JOINT_VEL_EQ_CNT : TMX_TERM_JOINT_VEL_INEQ_CNT;
      t.is_constraint = 1;
    }
    else
      return;
    t.first_step = first_step;
    t.last_step = last_step;
    for (std::size_t j = 0; j < n_dof; ++j)
    {
      t.coeffs[j] = coeffs[j];
      t.targets[j] = targets[j];
      t.upper_tols[j] = upper_tols[j];
      t.lower_tols[j] = lower_tols[j];
    }
    prob.addTerm(t, {}, name);
  }
};

/** problem_description.hpp:617-640 ; hatch: problem_description.cpp:1852-1890 - penalises sum_t dt_t - limit over the time variables of
    steps 1 .. n_steps - 1 (TimeCostCalculator / TimeCostJacCalculator, kinematic_terms.cpp:572-584); limit == 0 selects the SQUARED
    cost / EQ constraint, otherwise HINGE / INEQ */
struct TotalTimeTermInfo : public TermInfo
{
  double coeff{ 1.0 };
  double limit{ 1.0 };
  TotalTimeTermInfo() : TermInfo(TermType::TT_COST | TermType::TT_CNT | TermType::TT_USE_TIME) {}
  void hatch(TrajOptProb& prob) override
  {
    if (!prob.GetHasTime())
      printAndThrow(name + " uses time but the problem has no time variables (basic_info.use_time)");
    tmx_term t = detail::blankTerm();
    t.kind = TMX_TERM_TOTAL_TIME;
    t.first_step = 1;
    t.last_step = prob.GetNumSteps() - 1;
    t.coeff = coeff;
    t.margin = limit;
    if (static_cast<bool>(term_type & TermType::TT_COST))
      t.is_constraint = 0;
    else if (static_cast<bool>(term_type & TermType::TT_CNT))
      t.is_constraint = 1;
    else
      printAndThrow("A valid term type was not specified in TotalTimeTermInfo");  // :1886-1889
    prob.addTerm(t, {}, name);
  }
};

/** A function of the variables x[0 .. n_dof) of one waypoint as a tmx_expr stack program (include/tmx.h, interpreter
    include/tmx_expr.h): the device-evaluable stand-in for the reference's host callbacks sco::ScalarOfVector / VectorOfVector
    (trajopt_sco/include/trajopt_sco/num_diff.hpp:14-57), which cannot run inside a kernel.
        Expr x0 = Expr::var(0), x1 = Expr::var(1);   Expr f = sq(x1 - sq(x0)) + sq(1.0 - x0); */
class Expr
{
public:
  static Expr var(int i) { return Expr({ { TMX_OP_VAR, i } }, {}); }
  static Expr constant(double c) { return Expr({ { TMX_OP_CONST, 0 } }, { c }); }
  Expr(double c) : Expr(constant(c)) {}  // NOLINT(google-explicit-constructor): numbers mix with expressions
  friend Expr operator+(const Expr& a, const Expr& b) { return bin(a, b, TMX_OP_ADD); }
  friend Expr operator-(const Expr& a, const Expr& b) { return bin(a, b, TMX_OP_SUB); }
  friend Expr operator*(const Expr& a, const Expr& b) { return bin(a, b, TMX_OP_MUL); }
  friend Expr operator/(const Expr& a, const Expr& b) { return bin(a, b, TMX_OP_DIV); }
  Expr operator-() const { return un(*this, TMX_OP_NEG); }
  friend Expr sq(const Expr& a) { return un(a, TMX_OP_SQ); }
  friend Expr sin(const Expr& a) { return un(a, TMX_OP_SIN); }
  friend Expr cos(const Expr& a) { return un(a, TMX_OP_COS); }
  friend Expr sqrt(const Expr& a) { return un(a, TMX_OP_SQRT); }
  /** appends this expression followed by OUT(output) to a program */
  void emit(std::vector<int32_t>& ops, DblVec& consts, int output) const
  {
    std::size_t next_c = 0;
    for (const auto& oc : code_)
    {
      ops.push_back(oc.first);
      if (oc.first == TMX_OP_CONST)
      {
        consts.push_back(consts_[next_c++]);
        ops.push_back(static_cast<int32_t>(consts.size() - 1));
      }
      else
        ops.push_back(oc.second);
    }
    ops.push_back(TMX_OP_OUT);
    ops.push_back(output);
  }

private:
  using Code = std::vector<std::pair<int32_t, int32_t>>;
  Expr(Code code, DblVec consts) : code_(std::move(code)), consts_(std::move(consts)) {}
  static Expr bin(const Expr& a, const Expr& b, int32_t op)
  {
    Expr r = a;
    r.code_.insert(r.code_.end(), b.code_.begin(), b.code_.end());
    r.consts_.insert(r.consts_.end(), b.consts_.begin(), b.consts_.end());
    r.code_.emplace_back(op, 0);
    return r;
  }
  static Expr un(const Expr& a, int32_t op)
  {
    Expr r = a;
    r.code_.emplace_back(op, 0);
    return r;
  }
  Code code_;
  DblVec consts_;  // in the order of the CONST ops of code_
};

/** sco::CostFromFunc (trajopt_sco/src/modeling_utils.cpp:41-113) on the variables of every step in [first_step, last_step]; what a
    reference user writes as prob->addCost(std::make_shared<sco::CostFromFunc>(f, vars, name, full_hessian)) per step */
struct FuncCostTermInfo : public TermInfo
{
  Expr f{ 0.0 };
  int first_step = 0;
  int last_step = -1;
  bool full_hessian = false;
  FuncCostTermInfo() : TermInfo(TermType::TT_COST) {}
  void hatch(TrajOptProb& prob) override
  {
    tmx_term t = detail::blankTerm();
    t.kind = TMX_TERM_FUNC_COST;
    t.first_step = first_step;
    t.last_step = last_step <= -1 ? prob.GetNumSteps() - 1 : last_step;
    t.full_hessian = full_hessian ? 1 : 0;
    std::vector<int32_t> ops;
    DblVec consts;
    f.emit(ops, consts, 0);
    prob.addFuncTerm(t, ops, consts, 1, name);
  }
};

/** sco::ConstraintFromErrFunc without an analytic Jacobian (modeling_utils.cpp:213-269): g(x_t) == 0 (EQ) or <= 0 (INEQ), optional
    row coefficients */
struct FuncConstraintTermInfo : public TermInfo
{
  std::vector<Expr> g;
  int first_step = 0;
  int last_step = -1;
  bool ineq = false;
  DblVec coeffs;
  FuncConstraintTermInfo() : TermInfo(TermType::TT_CNT) {}
  void hatch(TrajOptProb& prob) override
  {
    if (g.empty() || g.size() > TMX_EXPR_MAX_OUT || (!coeffs.empty() && coeffs.size() != g.size()))
      printAndThrow("FuncConstraintTermInfo: 1 .. TMX_EXPR_MAX_OUT outputs, one coefficient per output if any");
    tmx_term t = detail::blankTerm();
    t.kind = TMX_TERM_FUNC_CNT;
    t.is_constraint = 1;
    t.first_step = first_step;
    t.last_step = last_step <= -1 ? prob.GetNumSteps() - 1 : last_step;
    t.cnt_type = ineq ? 1 : 0;
    t.has_coeffs = coeffs.empty() ? 0 : 1;
    for (std::size_t i = 0; i < coeffs.size(); ++i)
      t.coeffs[i] = coeffs[i];
    std::vector<int32_t> ops;
    DblVec consts;
    for (std::size_t i = 0; i < g.size(); ++i)
      g[i].emit(ops, consts, static_cast<int>(i));
    prob.addFuncTerm(t, ops, consts, static_cast<int>(g.size()), name);
  }
};

/** trajopt::UserDefinedTermInfo (problem_description.hpp:570-600 ; hatch: problem_description.cpp:599-675) with the error function
    given as Expr outputs instead of a host callback (numerical Jacobian, as the reference does without a jacobian_function):
    TT_COST -> TrajOptCostFromErrFunc with cost_penalty_type, TT_CNT -> TrajOptConstraintFromErrFunc with constraint_type; one term
    "name_<TYPE>_<step>" per step in [first_step, last_step] that is not in fixed_steps. */
struct UserDefinedTermInfo : public TermInfo
{
  enum PenaltyType
  {
    SQUARED = 0,
    ABS = 1,
    HINGE = 2
  };
  std::vector<Expr> error_function;
  int first_step = 0;
  int last_step = -1;
  DblVec coeff;
  PenaltyType cost_penalty_type = SQUARED;
  bool constraint_ineq = false;  // sco::ConstraintType: EQ (false) | INEQ
  IntVec fixed_steps;
  UserDefinedTermInfo() : TermInfo(TermType::TT_COST | TermType::TT_CNT) {}
  void hatch(TrajOptProb& prob) override
  {
    if (error_function.empty() || error_function.size() > TMX_EXPR_MAX_OUT || (!coeff.empty() && coeff.size() != error_function.size()))
      printAndThrow("UserDefinedTermInfo: 1 .. TMX_EXPR_MAX_OUT error outputs, one coefficient per output if any");
    tmx_term t = detail::blankTerm();
    const bool cnt = static_cast<bool>(term_type & TermType::TT_CNT) && !static_cast<bool>(term_type & TermType::TT_COST);
    t.kind = cnt ? TMX_TERM_FUNC_CNT : TMX_TERM_FUNC_ERR_COST;
    t.is_constraint = cnt ? 1 : 0;
    t.first_step = first_step;
    t.last_step = last_step <= -1 ? prob.GetNumSteps() - 1 : last_step;
    t.cnt_type = constraint_ineq ? 1 : 0;
    t.penalty_type = static_cast<int32_t>(cost_penalty_type);
    t.has_coeffs = coeff.empty() ? 0 : 1;
    for (std::size_t i = 0; i < coeff.size(); ++i)
      t.coeffs[i] = coeff[i];
    std::vector<int32_t> ops;
    DblVec consts;
    for (std::size_t i = 0; i < error_function.size(); ++i)
      error_function[i].emit(ops, consts, static_cast<int>(i));
    const std::string typ = cnt ? (constraint_ineq ? "INEQ" : "EQ") : (cost_penalty_type == ABS ? "ABS" : (cost_penalty_type == HINGE ? "HING" : "SQUARED"));
    prob.addFuncTerm(t, ops, consts, static_cast<int>(error_function.size()), name + "_" + typ, fixed_steps);
  }
};

/** JointAccTermInfo (problem_description.hpp:516-538 ; hatch: problem_description.cpp:1393-1493) and JointJerkTermInfo
    (:546-568 ; hatch :1515-1615): the four Eq / Ineq cost / constraint classes over the second / third difference
    (trajectory_costs.cpp:502-1016).  ORDER = 2 | 3. */
template <int ORDER>
struct JointDiffTermInfo : public TermInfo
{
  DblVec coeffs;
  DblVec targets;
  DblVec upper_tols;
  DblVec lower_tols;
  int first_step = 0;
  int last_step = -1;
  JointDiffTermInfo() : TermInfo(TermType::TT_COST | TermType::TT_CNT) {}
  void hatch(TrajOptProb& prob) override
  {
    const char* cls = ORDER == 2 ? "JointAccTermInfo" : "JointJerkTermInfo";
    const std::size_t n_dof = prob.GetKin()->numJoints();
    if (coeffs.empty())
      coeffs = DblVec(n_dof, 1);
    if (upper_tols.empty())
      upper_tols = DblVec(n_dof, 0);
    if (lower_tols.empty())
      lower_tols = DblVec(n_dof, 0);
    if (last_step <= -1)
      last_step = prob.GetNumSteps() - 1;
    if ((prob.GetNumSteps() - 1 - ORDER) <= first_step)  // :1407-1421 / :1529-1543
      first_step = prob.GetNumSteps() - 1 - ORDER;
    if ((prob.GetNumSteps() - 1) <= last_step)
      last_step = prob.GetNumSteps() - 1;
    if (last_step == first_step)
      last_step += (ORDER == 2 ? 2 : 4);  // (jerk: += 4 as the reference writes it, :1535)
    if (last_step < first_step)
      std::swap(first_step, last_step);
    detail::checkParameterSize(coeffs, n_dof, std::string(cls) + " coeffs");
    detail::checkParameterSize(targets, n_dof, std::string(cls) + " targets");
    detail::checkParameterSize(upper_tols, n_dof, std::string(cls) + " upper_tols");
    detail::checkParameterSize(lower_tols, n_dof, std::string(cls) + " lower_tols");
    if (first_step < 0 || last_step - ORDER - first_step < 0)
      printAndThrow(std::string(ORDER == 2 ? "JointAcc" : "JointJerk") + " term, trajectory is too short!");  // trajectory_costs.cpp:515, :768
    const bool zero = detail::allZero(upper_tols) && detail::allZero(lower_tols);
    tmx_term t = detail::blankTerm();
    if (static_cast<bool>(term_type & TermType::TT_USE_TIME))
      return;  // "Use time version of this term has not been defined." (:1439-1446): no term
    if (static_cast<bool>(term_type & TermType::TT_COST))
    {
      t.kind = ORDER == 2 ? (zero ? TMX_TERM_JOINT_ACC_EQ_COST : TMX_TERM_JOINT_ACC_INEQ_COST) :
                            (zero ? TMX_TERM_JOINT_JERK_EQ_COST : TMX_TERM_JOINT_JERK_INEQ_COST);
      t.is_constraint = 0;
    }
    else if (static_cast<bool>(term_type & TermType::TT_CNT))
    {
      t.kind = ORDER == 2 ? (zero ? TMX_TERM_JOINT_ACC_EQ_CNT : TMX_TERM_JOINT_ACC_INEQ_CNT) :
                            (zero ? TMX_TERM_JOINT_JERK_EQ_CNT : TMX_TERM_JOINT_JERK_INEQ_CNT);
      t.is_constraint = 1;
    }
    else
      return;
    t.first_step = first_step;
    t.last_step = last_step;
    for (std::size_t j = 0; j < n_dof; ++j)
    {
      t.coeffs[j] = coeffs[j];
      t.targets[j] = targets[j];
      t.upper_tols[j] = upper_tols[j];
      t.lower_tols[j] = lower_tols[j];
    }
    prob.addTerm(t, {}, name);
  }
};
using JointAccTermInfo = JointDiffTermInfo<2>;
using JointJerkTermInfo = JointDiffTermInfo<3>;

/** problem_description.hpp:352-392 ; hatch: problem_description.cpp:857-987.  Lowered: source = the manipulator's tip link
    (active), target = a static link of the environment; with a tolerance band the term runs on the dense QP engine. */
struct CartPoseTermInfo : public TermInfo
{
  int timestep{ 0 };
  std::string source_frame;
  std::string target_frame;
  Transform source_frame_offset;
  Transform target_frame_offset;
  std::array<double, 3> pos_coeffs{ { 1, 1, 1 } };
  std::array<double, 3> rot_coeffs{ { 1, 1, 1 } };
  DblVec lower_tolerance;
  DblVec upper_tolerance;
  CartPoseTermInfo() : TermInfo(TermType::TT_COST | TermType::TT_CNT) {}
  void hatch(TrajOptProb& prob) override
  {
    const auto kin = prob.GetKin();
    const auto env = prob.GetEnv();
    const bool src_active = (source_frame == kin->tip_link);
    const bool tgt_static = env && env->link_frames.count(target_frame) != 0;
    if (!src_active && !(env && env->link_frames.count(source_frame)))
      printAndThrow("invalid source frame: " + source_frame);  // :869
    if (!tgt_static && target_frame != kin->tip_link)
      printAndThrow("invalid target frame: " + target_frame);  // :874
    if (src_active && !tgt_static)
      printAndThrow("source '" + source_frame + "' and target '" + target_frame + "' are both active");  // :881
    if (!src_active && tgt_static)
      printAndThrow("source '" + source_frame + "' and target '" + target_frame + "' are both static");  // :886
    detail::validateTolerances("CartPoseErrCalculator", lower_tolerance, upper_tolerance);
    if (!source_frame_offset.isIdentity())
      printAndThrow("CartPoseTermInfo source_frame_offset: fold it into JointGroup::tool (one tool frame per problem)");
    if (timestep < 0 || timestep >= prob.GetNumSteps())
      printAndThrow("CartPoseTermInfo timestep out of range");
    tmx_term t = detail::blankTerm();
    t.kind = TMX_TERM_CART_POSE;
    t.first_step = t.last_step = timestep;
    for (std::size_t k = 0; k < 3; ++k)
    {
      t.coeffs[k] = pos_coeffs[k];
      t.coeffs[3 + k] = rot_coeffs[k];
    }
    const Transform target = env->link_frames.at(target_frame) * target_frame_offset;  // world_T_target * offset
    std::copy(target.m.begin(), target.m.end(), t.target_pose);
    detail::fillTolerances(t, lower_tolerance, upper_tolerance);
    if (static_cast<bool>(term_type & TermType::TT_COST))
      t.is_constraint = 0;  // ABS cost (:946-960)
    else if (static_cast<bool>(term_type & TermType::TT_CNT))
      t.is_constraint = 1;  // EQ constraint (:961-976)
    else
      return;
    prob.addTerm(t, {}, name);
  }
};

/** trajopt::DynamicCartPoseTermInfo (problem_description.hpp:310-350; hatch problem_description.cpp:752-822): source and target are
    BOTH links of the manipulator.  Lowered: source = the tip link (tool frame), target = any moving link times
    target_frame_offset. */
struct DynamicCartPoseTermInfo : public TermInfo
{
  int timestep{ 0 };
  std::string source_frame;
  std::string target_frame;
  Transform source_frame_offset;
  Transform target_frame_offset;
  std::array<double, 3> pos_coeffs{ { 1, 1, 1 } };
  std::array<double, 3> rot_coeffs{ { 1, 1, 1 } };
  DblVec lower_tolerance;
  DblVec upper_tolerance;
  DynamicCartPoseTermInfo() : TermInfo(TermType::TT_COST | TermType::TT_CNT) {}
  void hatch(TrajOptProb& prob) override
  {
    const auto kin = prob.GetKin();
    const int target = kin->linkIndex(target_frame);
    if (source_frame != kin->tip_link || target < 0)
      printAndThrow("source '" + source_frame + "' and target '" + target_frame + "' are not both active links");  // :735
    detail::validateTolerances("DynamicCartPoseErrCalculator", lower_tolerance, upper_tolerance);
    if (!source_frame_offset.isIdentity())
      printAndThrow("DynamicCartPoseTermInfo source_frame_offset: fold it into JointGroup::tool (one tool frame per problem)");
    if (timestep < 0 || timestep >= prob.GetNumSteps())
      printAndThrow("DynamicCartPoseTermInfo timestep out of range");
    if (static_cast<bool>(term_type & TermType::TT_USE_TIME))
      printAndThrow("Use time version of this term has not been defined.");  // :781-784
    tmx_term t = detail::blankTerm();
    t.kind = TMX_TERM_DYN_CART_POSE;
    t.first_step = t.last_step = timestep;
    for (std::size_t k = 0; k < 3; ++k)
    {
      t.coeffs[k] = pos_coeffs[k];
      t.coeffs[3 + k] = rot_coeffs[k];
    }
    t.link = target;
    std::copy(target_frame_offset.m.begin(), target_frame_offset.m.end(), t.target_pose);
    detail::fillTolerances(t, lower_tolerance, upper_tolerance);
    if (static_cast<bool>(term_type & TermType::TT_COST))
      t.is_constraint = 0;  // ABS cost (:806-810)
    else if (static_cast<bool>(term_type & TermType::TT_CNT))
      t.is_constraint = 1;  // EQ constraint (:811-815)
    else
      return;
    prob.addTerm(t, {}, name);
  }
};

/** trajopt::AvoidSingularityTermInfo (problem_description.hpp:637-659; hatch problem_description.cpp:1900-1940) over the problem's
    joint set or a subset of it: one ABS cost / INEQ constraint per step in [first_step, last_step] */
struct AvoidSingularityTermInfo : public TermInfo
{
  /** optional joint subset (problem_description.hpp:640): a group whose joints are a run of the manipulator's joints ending at the
      joint of `link`; used when its joint names are a subset of the problem's (problem_description.cpp:1907) */
  std::shared_ptr<const JointGroup> subset_kin_;
  double lambda{ 0.1 };
  std::string link;
  int first_step{ -1 };
  int last_step{ -1 };
  DblVec coeffs;
  explicit AvoidSingularityTermInfo(double lambda_ = 0.1) : TermInfo(TermType::TT_COST | TermType::TT_CNT), lambda(lambda_) {}
  void hatch(TrajOptProb& prob) override
  {
    const auto kin = prob.GetKin();
    const int idx = kin->linkIndex(link);
    if (idx < 0)
      printAndThrow("invalid link name: " + link);
    if (first_step < 0 || last_step >= prob.GetNumSteps() || first_step > last_step)
      printAndThrow("avoid_singularity: first_step / last_step out of range");
    if (coeffs.size() != 1)
      printAndThrow("avoid_singularity: one coefficient (the error has one row)");
    tmx_term t = detail::blankTerm();
    t.kind = TMX_TERM_AVOID_SINGULARITY;
    t.first_step = first_step;
    t.last_step = last_step;
    t.coeffs[0] = coeffs[0];
    t.link = idx;
    t.lambda = lambda;
    if (subset_kin_ && !subset_kin_->joint_names.empty())
    {
      // isSuperset(subset names, problem names) (:1907); lowered when the subset is the run of joints first .. link
      const auto& all = kin->joint_names;
      const auto first = std::find(all.begin(), all.end(), subset_kin_->joint_names.front());
      bool subset = true;
      for (const auto& n : subset_kin_->joint_names)
        subset = subset && std::find(all.begin(), all.end(), n) != all.end();
      if (subset)
      {
        const int j0 = static_cast<int>(first - all.begin());
        if (j0 + static_cast<int>(subset_kin_->joint_names.size()) != idx + 1 ||
            !std::equal(subset_kin_->joint_names.begin(), subset_kin_->joint_names.end(), first))
          printAndThrow("avoid_singularity: the joint subset must be a run of the manipulator's joints that ends at the link's joint");
        t.subset_first = j0 + 1;
      }
    }
    if (static_cast<bool>(term_type & TermType::TT_COST))
      t.is_constraint = 0;  // ABS cost (:1925-1929)
    else if (static_cast<bool>(term_type & TermType::TT_CNT))
      t.is_constraint = 1;  // INEQ constraint (:1930-1934)
    else
      return;
    prob.addTerm(t, {}, name);
  }
};

/** trajopt::CartVelTermInfo (problem_description.hpp:394-420, problem_description.cpp:989-1057): the tool-frame origin may move at
    most max_displacement per axis between waypoints i and i + 1, i in [first_step, last_step] */
struct CartVelTermInfo : public TermInfo
{
  int first_step{ 0 };
  int last_step{ 0 };
  double max_displacement{ 0.0 };
  std::string link;
  CartVelTermInfo() : TermInfo(TermType::TT_COST | TermType::TT_CNT) {}
  void hatch(TrajOptProb& prob) override
  {
    const auto kin = prob.GetKin();
    const int n_steps = prob.GetNumSteps();
    if (link != kin->tip_link)
      printAndThrow("invalid link name: " + link);  // :1003 (only the tip link carries the tool frame here)
    if (!((first_step >= 0) && (first_step <= n_steps - 1) && (first_step < last_step)) || !((last_step > 0) && (last_step <= n_steps - 1)))
      printAndThrow("cart_vel: first_step / last_step out of range");  // FAIL_IF_FALSE :997-998
    if (last_step + 1 > n_steps - 1)
      printAndThrow("cart_vel: last_step + 1 must be a waypoint of the trajectory (the term couples steps i and i + 1)");
    if (static_cast<bool>(term_type & TermType::TT_USE_TIME))
      printAndThrow("Use time version of this term has not been defined.");  // :1015-1022
    tmx_term t = detail::blankTerm();
    t.kind = TMX_TERM_CART_VEL;
    t.first_step = first_step;
    t.last_step = last_step;
    t.margin = max_displacement;
    if (static_cast<bool>(term_type & TermType::TT_COST))
      t.is_constraint = 0;  // ABS cost (:1023-1036)
    else if (static_cast<bool>(term_type & TermType::TT_CNT))
      t.is_constraint = 1;  // INEQ constraint (:1038-1052)
    else
      return;
    prob.addTerm(t, {}, name);
  }
};

/** the members of trajopt_common::TrajOptCollisionConfig (trajopt_common/include/trajopt_common/collision_types.h:119-162)
    that the collision terms read */
struct TrajOptCollisionConfig
{
  enum class CollisionEvaluatorType  // tesseract::collision::CollisionEvaluatorType
  {
    NONE = 0,
    DISCRETE = 1,
    LVS_DISCRETE = 2,
    CONTINUOUS = 3,
    LVS_CONTINUOUS = 4
  };
  bool enabled{ true };
  double default_margin{ 0.0 };           // contact_manager_config.default_margin (JSON "dist_pen")
  double default_collision_coeff{ 1.0 };  // collision_coeff_data default (JSON "coeffs")
  double collision_margin_buffer{ 0.01 };
  CollisionEvaluatorType type{ CollisionEvaluatorType::DISCRETE };  // collision_check_config.type
  double longest_valid_segment_length{ 0.005 };                     // collision_check_config.longest_valid_segment_length
  /** row-slot capacity per (segment, link sphere, obstacle) of the device path for the LVS / continuous evaluators: the
      number of sub-states ceil(dist / lvs) + 1 is clamped to it (not a reference member) */
  int max_substates{ 2 };
  TrajOptCollisionConfig() = default;
  TrajOptCollisionConfig(double margin, double coeff) : default_margin(margin), default_collision_coeff(coeff) {}
};

/** problem_description.hpp:597-617 ; hatch: problem_description.cpp:1716-1837.  DISCRETE evaluator: one term per non-fixed
    step (SINGLE_TIME_STEP expressions); LVS_DISCRETE / CONTINUOUS / LVS_CONTINUOUS: one term per segment (i, i+1) with
    START_FREE_END_FREE / START_FIXED_END_FREE / START_FREE_END_FIXED expressions; cost and constraint form. */
struct CollisionTermInfo : public TermInfo
{
  int first_step{ -1 }, last_step{ -1 };
  std::vector<int> fixed_steps;
  TrajOptCollisionConfig config;
  CollisionTermInfo() : TermInfo(TermType::TT_COST | TermType::TT_CNT) {}
  void hatch(TrajOptProb& prob) override
  {
    if (!config.enabled)
      return;
    if (config.type == TrajOptCollisionConfig::CollisionEvaluatorType::NONE)
      printAndThrow("CollisionTermInfo: evaluator type NONE");
    if (first_step < 0 || last_step < first_step || last_step >= prob.GetNumSteps())
      printAndThrow("CollisionTermInfo: invalid first_step / last_step");
    for (int fs : fixed_steps)
      if (fs < first_step || fs > last_step)
        printAndThrow("Fixed step " + std::to_string(fs) + " is not between first step " + std::to_string(first_step) + " and last step " +
                      std::to_string(last_step));  // :1645
    tmx_term t = detail::blankTerm();
    t.first_step = first_step;
    t.last_step = last_step;
    t.margin = config.default_margin;
    t.coeff = config.default_collision_coeff;
    t.buffer = config.collision_margin_buffer;
    t.evaluator_type = static_cast<int32_t>(config.type);
    t.longest_valid_segment_length = config.longest_valid_segment_length;
    t.max_substates = config.max_substates;
    if (t.evaluator_type >= 2 && t.max_substates <= 0)
    {
      // capacity from the initial trajectory: 1.5 x the longest segment, at most 64 sub-states
      const TrajArray& it = prob.GetInitTraj();
      double dmax = 0.0;
      for (int i = 0; i + 1 < it.rows(); ++i)
      {
        double d2 = 0.0;
        for (int j = 0; j < it.cols(); ++j)
          d2 += (it(i + 1, j) - it(i, j)) * (it(i + 1, j) - it(i, j));
        dmax = std::max(dmax, std::sqrt(d2));
      }
      const double lv = std::max(config.longest_valid_segment_length, 1e-9);
      t.max_substates = static_cast<int32_t>(std::min(64.0, std::max(2.0, std::ceil(1.5 * dmax / lv) + 1.0)));
    }
    if (static_cast<bool>(term_type & TermType::TT_COST))
    {
      t.kind = TMX_TERM_COLLISION_COST;
      t.is_constraint = 0;
    }
    else
    {
      t.kind = TMX_TERM_COLLISION_CNT;
      t.is_constraint = 1;
    }
    prob.addTerm(t, fixed_steps, name);
  }
};

namespace detail
{
/** generateInitTraj, problem_description.cpp:310-366 */
inline TrajArray generateInitTraj(const ProblemConstructionInfo& pci)
{
  const InitInfo& init_info = pci.init_info;
  const int n_steps = pci.basic_info.n_steps;
  const int D = static_cast<int>(pci.kin->numJoints());
  auto current = [&]() {
    auto it = pci.env->state.find(pci.basic_info.manip);
    DblVec s = (it == pci.env->state.end()) ? DblVec(static_cast<std::size_t>(D), 0.0) : it->second;
    if (static_cast<int>(s.size()) != D)
      printAndThrow("environment state has the wrong number of joint values for " + pci.basic_info.manip);
    return s;
  };
  TrajArray init;
  if (init_info.type == InitInfo::STATIONARY)
  {
    const DblVec s = current();
    init = TrajArray(n_steps, D);
    for (int t = 0; t < n_steps; ++t)
      for (int j = 0; j < D; ++j)
        init(t, j) = s[static_cast<std::size_t>(j)];
  }
  else if (init_info.type == InitInfo::JOINT_INTERPOLATED)
  {
    const DblVec s = current();
    const TrajArray& d = init_info.data;
    if (!((d.rows() == 1 && d.cols() == D) || (d.rows() == D && d.cols() == 1)))
      printAndThrow("JOINT_INTERPOLATED selected, but init_info.data is the wrong size. It should be 1 x pci.kin->numJoints()");
    init = TrajArray(n_steps, D);
    for (int j = 0; j < D; ++j)
    {
      // Eigen::VectorXd::LinSpaced(n_steps, start, end): start + i*step, the last entry exactly `end`
      const double a = s[static_cast<std::size_t>(j)], b = d.data[static_cast<std::size_t>(j)];
      const double step = n_steps > 1 ? (b - a) / (n_steps - 1) : 0.0;
      for (int t = 0; t < n_steps; ++t)
        init(t, j) = (t == n_steps - 1 && n_steps > 1) ? b : a + t * step;
    }
  }
  else if (init_info.type == InitInfo::GIVEN_TRAJ)
    init = init_info.data;
  else
    printAndThrow("Init Info did not have a valid type. Valid types are STATIONARY, JOINT_INTERPOLATED, or GIVEN_TRAJ");
  return init;
}
}  // namespace detail

/** ConstructProblem(const ProblemConstructionInfo&), problem_description.cpp:410-592 */
inline TrajOptProb::Ptr ConstructProblem(const ProblemConstructionInfo& pci)
{
  const BasicInfo& bi = pci.basic_info;
  const int n_steps = bi.n_steps;
  if (!pci.env || !pci.kin)
    printAndThrow("ProblemConstructionInfo needs env and kin");
  if (n_steps < 1)
    printAndThrow("basic_info.n_steps must be positive");
  // term-type checks, :417-453
  bool use_time = false;
  for (const TermInfo::Ptr& cost : pci.cost_infos)
  {
    if (!static_cast<bool>(cost->getSupportedTypes() & TermType::TT_COST))
      printAndThrow(cost->name + " is only a constraint, but you listed it as a cost");
    if (static_cast<bool>(cost->term_type & TermType::TT_USE_TIME))
    {
      use_time = true;
      if (!static_cast<bool>(cost->getSupportedTypes() & TermType::TT_USE_TIME))
        printAndThrow(cost->name + " does not support time, but you listed it as a using time");
    }
  }
  for (const TermInfo::Ptr& cnt : pci.cnt_infos)
  {
    if (!static_cast<bool>(cnt->getSupportedTypes() & TermType::TT_CNT))
      printAndThrow(cnt->name + " is only a cost, but you listed it as a constraint");
    if (static_cast<bool>(cnt->term_type & TermType::TT_USE_TIME))
    {
      use_time = true;
      if (!static_cast<bool>(cnt->getSupportedTypes() & TermType::TT_USE_TIME))
        printAndThrow(cnt->name + " does not support time, but you listed it as a using time");
    }
  }
  if (use_time && !bi.use_time)  // :447-452
    printAndThrow("A term is using time and basic_info is not set correctly. Try basic_info.use_time = true");
  if (!use_time && bi.use_time)
    printAndThrow("No terms use time and basic_info is not set correctly. Try basic_info.use_time = false");
  if (bi.dt_lower_lim <= 0 || bi.dt_upper_lim < bi.dt_lower_lim)  // readBasicInfo :129-133
    printAndThrow("dt limits (Basic Info) invalid. The lower limit must be positive, and the minimum upper limit is equal to the lower limit.");

  auto prob = std::make_shared<TrajOptProb>(n_steps, pci);
  const int n_dof = prob->GetNumDOF();
  TrajArray init_traj = detail::generateInitTraj(pci);
  if (bi.use_time)
  {
    // "Currently all trajectories are generated without time then appended here" (:367-376): the time column is init_info.dt
    if (n_dof + 1 > TMX_MAX_DOF)
      printAndThrow("n_dof + 1 (time column) exceeds TMX_MAX_DOF");
    prob->SetHasTime(true);
    prob->setTimeLimits(bi.dt_lower_lim, bi.dt_upper_lim);
    TrajArray with_time(init_traj.rows(), init_traj.cols() + 1);
    for (int t = 0; t < init_traj.rows(); ++t)
    {
      for (int j = 0; j < init_traj.cols(); ++j)
        with_time(t, j) = init_traj(t, j);
      with_time(t, init_traj.cols()) = pci.init_info.dt;
    }
    init_traj = with_time;
  }
  const int n_cols = n_dof + (bi.use_time ? 1 : 0);
  if (init_traj.rows() != n_steps || init_traj.cols() != n_cols)  // :460-481
    printAndThrow("Initial trajectory is not the right size matrix\nExpected " + std::to_string(n_steps) + " rows (time steps) x " +
                  std::to_string(n_cols) + " columns\nGot " + std::to_string(init_traj.rows()) + " rows and " +
                  std::to_string(init_traj.cols()) + " columns");
  prob->SetInitTraj(init_traj);
  for (const int t_idx : bi.fixed_timesteps)  // :485-508
    if (t_idx < 0 || t_idx >= init_traj.rows())
      printAndThrow("Fixed timestep index is outside the bounds of the initial trajectory.");
  for (const int dof_ind : bi.fixed_dofs)  // :510-530
    if (dof_ind < 0 || dof_ind >= n_dof)
      printAndThrow("DOF(aka Joint) indice is greater than the number of DOF available.");
  prob->setFixed(bi.fixed_timesteps, bi.fixed_dofs);
  // costs first, then constraints (:560-571).  A TermInfo listed under cost_infos / cnt_infos is hatched as that kind,
  // whatever its term_type says (the reference only warns, :420-421 / :434-435).
  for (const TermInfo::Ptr& ci : pci.cost_infos)
  {
    ci->term_type = static_cast<bool>(ci->term_type & TermType::TT_USE_TIME) ? (TermType::TT_COST | TermType::TT_USE_TIME) : TermType::TT_COST;
    ci->hatch(*prob);
  }
  for (const TermInfo::Ptr& ci : pci.cnt_infos)
  {
    ci->term_type = static_cast<bool>(ci->term_type & TermType::TT_USE_TIME) ? (TermType::TT_CNT | TermType::TT_USE_TIME) : TermType::TT_CNT;
    ci->hatch(*prob);
  }
  return prob;
}
}  // namespace trajopt

namespace sco
{
/** optimizers.hpp:230-316 BasicTrustRegionSQPResults: what one trust-region evaluation (Model::optimize + exact re-evaluation)
    leaves for the per-iteration table and the log writers.  Filled from tmx_sqp_step_log; model_var_vals / new_x are not
    carried (new_x is OptResults::x after an accepted step). */
struct BasicTrustRegionSQPResults
{
  std::size_t seed{ 0 };
  int merit_increases{ 0 }, sqp_iter{ 0 };
  double trust_box_size{ 0 };  ///< the box the QP of this evaluation was solved with
  DblVec model_cost_vals, model_cnt_viols, new_cost_vals, old_cost_vals, new_cnt_viols, old_cnt_viols;
  double old_merit{ 0 }, model_merit{ 0 }, new_merit{ 0 };
  double approx_merit_improve{ 0 }, exact_merit_improve{ 0 }, merit_improve_ratio{ 0 };
  std::vector<double> merit_error_coeffs;
  std::vector<std::string> cost_names, cnt_names;

  /** BasicTrustRegionSQPResults::print (optimizers.cpp:428-531), same columns and formats, into a stream */
  void print(std::FILE* out = stdout) const
  {
    const std::string bar(88, '='), dash(88, '-');
    std::fprintf(out, "\n| %s |\n", bar.c_str());
    std::fprintf(out, "| %10s | %10s | %10s | %10s | %10s | %10s | %10s |\n", "merit", "oldexact", "new_exact", "new_approx", "dapprox", "dexact",
                 "ratio");
    std::fprintf(out, "| %s | COSTS\n", dash.c_str());
    auto name = [](const std::vector<std::string>& n, std::size_t i) { return i < n.size() ? n[i].c_str() : ""; };
    auto sum = [](const DblVec& v) {
      double o = 0;
      for (double x : v)
        o += x;
      return o;
    };
    for (std::size_t i = 0; i < old_cost_vals.size(); ++i)
    {
      const double approx_improve = old_cost_vals[i] - model_cost_vals[i];
      const double exact_improve = old_cost_vals[i] - new_cost_vals[i];
      if (std::fabs(approx_improve) > 1e-8)
        std::fprintf(out, "| %10s | %10.3e | %10.3e | %10.3e | %10.3e | %10.3e | %10.3e | %-15s \n", "----------", old_cost_vals[i],
                     new_cost_vals[i], model_cost_vals[i], approx_improve, exact_improve, exact_improve / approx_improve, name(cost_names, i));
      else
        std::fprintf(out, "| %10s | %10.3e | %10.3e | %10.3e | %10.3e | %10.3e | %10s | %-15s \n", "----------", old_cost_vals[i],
                     new_cost_vals[i], model_cost_vals[i], approx_improve, exact_improve, "----------", name(cost_names, i));
    }
    std::fprintf(out, "| %s |\n", bar.c_str());
    std::fprintf(out, "| %10s | %10.3e | %10.3e | %10.3e | %10s | %10s | %10s | SUM COSTS\n", "----------", sum(old_cost_vals),
                 sum(new_cost_vals), sum(model_cost_vals), "----------", "----------", "----------");
    std::fprintf(out, "| %s |\n", bar.c_str());
    if (!old_cnt_viols.empty())
    {
      std::fprintf(out, "| %s | CONSTRAINTS\n", dash.c_str());
      for (std::size_t i = 0; i < old_cnt_viols.size(); ++i)
      {
        const double approx_improve = old_cnt_viols[i] - model_cnt_viols[i];
        const double exact_improve = old_cnt_viols[i] - new_cnt_viols[i];
        const double mc = merit_error_coeffs[i];
        if (std::fabs(approx_improve) > 1e-8)
          std::fprintf(out, "| %10.3e | %10.3e | %10.3e | %10.3e | %10.3e | %10.3e | %10.3e | %-15s \n", mc, mc * old_cnt_viols[i],
                       mc * new_cnt_viols[i], mc * model_cnt_viols[i], mc * approx_improve, mc * exact_improve, exact_improve / approx_improve,
                       name(cnt_names, i));
        else
          std::fprintf(out, "| %10.3e | %10.3e | %10.3e | %10.3e | %10.3e | %10.3e | %10s | %-15s \n", mc, mc * old_cnt_viols[i],
                       mc * new_cnt_viols[i], mc * model_cnt_viols[i], mc * approx_improve, mc * exact_improve, "----------", name(cnt_names, i));
      }
    }
    std::fprintf(out, "| %s |\n", bar.c_str());
    std::fprintf(out, "| %10s | %10.3e | %10.3e | %10.3e | %10s | %10s | %10s | SUM CONSTRAINTS (WITHOUT MERIT) \n", "----------",
                 sum(old_cnt_viols), sum(new_cnt_viols), sum(model_cnt_viols), "----------", "----------", "----------");
    std::fprintf(out, "| %s |\n", bar.c_str());
    std::fprintf(out, "| %10s | %10.3e | %10.3e | %10s | %10.3e | %10.3e | %10.3e | TOTAL = SUM COSTS + SUM CONSTRAINTS (WITH MERIT)\n",
                 "----------", old_merit, new_merit, "----------", approx_merit_improve, exact_merit_improve, merit_improve_ratio);
    std::fprintf(out, "| %s |\n", bar.c_str());
  }
};

/** Sibling of sco::BasicTrustRegionSQPMultiThreaded (optimizers.hpp:196-218): BasicTrustRegionSQP::optimize() for a batch
    of seeds on one MI355X.  Throws if no device is available — there is no CPU path behind it. */
class BasicTrustRegionSQPBatchedHip
{
public:
  using Callback = std::function<void(trajopt::TrajOptProb*, OptResults&)>;  // optimizers.hpp:83-84

  explicit BasicTrustRegionSQPBatchedHip(trajopt::TrajOptProb::Ptr prob, int device = 0) : prob_(std::move(prob))
  {
    if (!prob_)
      printAndThrow("need to set the problem before initializing");
    const tmx_status rc = tmx_create(device, &ctx_);
    if (rc != TMX_OK || !ctx_)
      printAndThrow("tmx_create failed: no MI355X device available (libtrajopt_mi355x has no CPU fallback)");
    tmx_default_osqp_settings(&osqp_);
  }
  ~BasicTrustRegionSQPBatchedHip()
  {
    if (ctx_)
      tmx_destroy(ctx_);
  }
  BasicTrustRegionSQPBatchedHip(const BasicTrustRegionSQPBatchedHip&) = delete;
  BasicTrustRegionSQPBatchedHip& operator=(const BasicTrustRegionSQPBatchedHip&) = delete;

  void setParameters(const BasicTrustRegionSQPParameters& param) { param_ = param; }
  const BasicTrustRegionSQPParameters& getParameters() const { return param_; }
  BasicTrustRegionSQPParameters& getParameters() { return param_; }
  /** OSQPModelConfig::settings (osqp_interface.cpp:78-90 defaults) */
  tmx_osqp_settings& getOSQPSettings() { return osqp_; }
  void addCallback(const Callback& cb) { callbacks_.push_back(cb); }  // invoked once per optimize(), on the best seed
  /** the loop variables the reference's per-iteration log shows (optimizers.cpp:742-760) */
  struct IterationInfo
  {
    std::size_t seed{ 0 };
    int sqp_iter{ 0 };
    int merit_increases{ 0 };
    double trust_box_size{ 0 };
    bool at_exit{ false };
  };
  using IterationCallback = std::function<void(const IterationInfo&, OptResults&)>;
  /** Per-iteration observability: the callback runs before every SQP iteration of every seed and once when the seed
      finishes, as Optimizer callbacks do in the reference (optimizers.cpp:754, :978).  With such a callback the batch
      is stepped one trust-region evaluation per launch (tmx_sqp_run(max_steps = 1)) instead of running in the
      persistent kernel; the results are identical. */
  void addIterationCallback(const IterationCallback& cb) { iteration_callbacks_.push_back(cb); }
  /** After EVERY trust-region evaluation of every seed: the BasicTrustRegionSQPResults the reference prints / logs per
      iteration (optimizers.cpp:380-647).  Also puts the optimizer into the stepped mode. */
  using ResultsCallback = std::function<void(const BasicTrustRegionSQPResults&)>;
  void addResultsCallback(const ResultsCallback& cb) { results_callbacks_.push_back(cb); }

  /** Optimizer::initialize (optimizers.cpp:127-136): one seed, x in j_t_d order (row-major n_steps x n_dof) */
  void initialize(const DblVec& x) { initialize(std::vector<DblVec>{ x }); }
  /** a batch of seeds */
  void initialize(const std::vector<DblVec>& seeds)
  {
    const std::size_t nv = numVars();
    if (seeds.empty())
      printAndThrow("initialize: empty batch");
    for (const DblVec& x : seeds)
      if (x.size() != nv)
        printAndThrow("initialization vector has wrong length. expected " + std::to_string(nv) + " got " + std::to_string(x.size()));
    seeds_.clear();
    seeds_.reserve(seeds.size() * nv);
    for (const DblVec& x : seeds)
      seeds_.insert(seeds_.end(), x.begin(), x.end());
    batch_ = static_cast<int32_t>(seeds.size());
    results_.clear();
    results_.x = seeds[0];
    batch_results_.clear();
  }

  /** BasicTrustRegionSQP::optimize() (optimizers.cpp:699-991) for every seed; returns the status of the best seed */
  OptStatus optimize()
  {
    if (batch_ == 0)
      printAndThrow("must initialize before optimizing");
    const tmx_sqp_params p = param_.toTmx();
    check(tmx_problem_upload(ctx_, &prob_->desc(), &p, &osqp_));
    check(tmx_batch_set_x0(ctx_, seeds_.data(), batch_));
    if (iteration_callbacks_.empty() && results_callbacks_.empty())
      check(tmx_sqp_run(ctx_, 0, nullptr));
    else
      runStepwise();
    const std::size_t B = static_cast<std::size_t>(batch_), nv = numVars();
    std::vector<double> x(B * nv), cost(B);
    std::vector<int32_t> status(B), nfe(B), nqp(B);
    check(tmx_sqp_results(ctx_, x.data(), status.data(), cost.data(), nfe.data(), nqp.data()));
    int32_t n_costs = 0, n_cnts = 0, n_slots = 0;
    check(tmx_term_counts(ctx_, &n_costs, &n_cnts, &n_slots));
    std::vector<double> cv(B * static_cast<std::size_t>(n_costs)), vv(B * static_cast<std::size_t>(n_cnts));
    check(tmx_evaluate(ctx_, cv.data(), vv.data()));  // cost values / constraint violations at the final iterates
    batch_results_.assign(B, OptResults());
    for (std::size_t b = 0; b < B; ++b)
    {
      OptResults& r = batch_results_[b];
      r.x.assign(x.begin() + static_cast<std::ptrdiff_t>(b * nv), x.begin() + static_cast<std::ptrdiff_t>((b + 1) * nv));
      r.status = static_cast<OptStatus>(status[b]);
      r.total_cost = cost[b];
      r.n_func_evals = nfe[b];
      r.n_qp_solves = nqp[b];
      r.cost_vals.assign(cv.begin() + static_cast<std::ptrdiff_t>(b * static_cast<std::size_t>(n_costs)),
                         cv.begin() + static_cast<std::ptrdiff_t>((b + 1) * static_cast<std::size_t>(n_costs)));
      r.cnt_viols.assign(vv.begin() + static_cast<std::ptrdiff_t>(b * static_cast<std::size_t>(n_cnts)),
                         vv.begin() + static_cast<std::ptrdiff_t>((b + 1) * static_cast<std::size_t>(n_cnts)));
    }
    int64_t best = -1;
    double best_cost = 0;
    check(tmx_argmin(ctx_, 0, &best, &best_cost));  // lowest total_cost among the converged seeds
    best_ = best < 0 ? 0 : static_cast<std::size_t>(best);
    results_ = batch_results_[best_];
    for (auto& cb : callbacks_)
      cb(prob_.get(), results_);
    return results_.status;
  }

  DblVec& x() { return results_.x; }
  OptResults& results() { return results_; }
  const std::vector<OptResults>& batchResults() const { return batch_results_; }
  std::size_t bestSeed() const { return best_; }
  tmx_ctx* context() { return ctx_; }

private:
  void runStepwise()
  {
    const std::size_t B = static_cast<std::size_t>(batch_), nv = numVars();
    int32_t n_costs = 0, n_cnts = 0, n_slots = 0;
    std::vector<long long> seen(B, -1);
    std::vector<char> finished(B, 0);
    std::vector<double> x(B * nv), cost(B), trust(B);
    std::vector<int32_t> status(B), nfe(B), nqp(B), it(B), mi(B), done(B), nqp_seen(B, 0);
    while (true)
    {
      check(tmx_sqp_results(ctx_, x.data(), status.data(), cost.data(), nfe.data(), nqp.data()));
      check(tmx_sqp_state(ctx_, it.data(), mi.data(), trust.data(), done.data()));
      check(tmx_term_counts(ctx_, &n_costs, &n_cnts, &n_slots));
      if (!results_callbacks_.empty())
      {
        int32_t stride = 0;
        check(tmx_sqp_step_log(ctx_, nullptr, &stride));
        std::vector<double> lg(B * static_cast<std::size_t>(stride));
        check(tmx_sqp_step_log(ctx_, lg.data(), &stride));
        const std::size_t nc = static_cast<std::size_t>(n_costs), nv2 = static_cast<std::size_t>(n_cnts);
        for (std::size_t b = 0; b < B; ++b)
        {
          const double* o = lg.data() + b * static_cast<std::size_t>(stride);
          if (nqp[b] <= nqp_seen[b] || o[9] == 0.0)
          {
            nqp_seen[b] = nqp[b];
            continue;
          }
          nqp_seen[b] = nqp[b];
          BasicTrustRegionSQPResults r;
          r.seed = b;
          r.merit_increases = static_cast<int>(o[0]);
          r.sqp_iter = static_cast<int>(o[1]);
          r.trust_box_size = o[2];
          r.old_merit = o[3];
          r.model_merit = o[4];
          r.new_merit = o[5];
          r.approx_merit_improve = o[6];
          r.exact_merit_improve = o[7];
          r.merit_improve_ratio = o[8];
          const double* q = o + TMX_STEP_LOG_HEAD;
          r.old_cost_vals.assign(q, q + nc);
          r.model_cost_vals.assign(q + nc, q + 2 * nc);
          r.new_cost_vals.assign(q + 2 * nc, q + 3 * nc);
          q += 3 * nc;
          r.old_cnt_viols.assign(q, q + nv2);
          r.model_cnt_viols.assign(q + nv2, q + 2 * nv2);
          r.new_cnt_viols.assign(q + 2 * nv2, q + 3 * nv2);
          r.merit_error_coeffs.assign(q + 3 * nv2, q + 4 * nv2);
          r.cost_names = prob_->getCostNames();
          r.cnt_names = prob_->getCntNames();
          for (auto& cb : results_callbacks_)
            cb(r);
        }
      }
      std::vector<double> cv(B * static_cast<std::size_t>(n_costs)), vv(B * static_cast<std::size_t>(n_cnts));
      check(tmx_evaluate(ctx_, cv.data(), vv.data()));
      bool all = true;
      for (std::size_t b = 0; b < B; ++b)
      {
        if (finished[b])
          continue;
        const long long key = static_cast<long long>(mi[b]) * 100000 + it[b];
        const bool fire = done[b] || key != seen[b];
        if (done[b])
          finished[b] = 1;
        else
        {
          seen[b] = key;
          all = false;
        }
        if (!fire)
          continue;
        OptResults r;
        r.x.assign(x.begin() + static_cast<std::ptrdiff_t>(b * nv), x.begin() + static_cast<std::ptrdiff_t>((b + 1) * nv));
        r.status = static_cast<OptStatus>(status[b]);
        r.total_cost = cost[b];
        r.n_func_evals = nfe[b];
        r.n_qp_solves = nqp[b];
        r.cost_vals.assign(cv.begin() + static_cast<std::ptrdiff_t>(b * static_cast<std::size_t>(n_costs)),
                           cv.begin() + static_cast<std::ptrdiff_t>((b + 1) * static_cast<std::size_t>(n_costs)));
        r.cnt_viols.assign(vv.begin() + static_cast<std::ptrdiff_t>(b * static_cast<std::size_t>(n_cnts)),
                           vv.begin() + static_cast<std::ptrdiff_t>((b + 1) * static_cast<std::size_t>(n_cnts)));
        IterationInfo info;
        info.seed = b;
        info.sqp_iter = it[b];
        info.merit_increases = mi[b];
        info.trust_box_size = trust[b];
        info.at_exit = done[b] != 0;
        for (auto& cb : iteration_callbacks_)
          cb(info, r);
      }
      if (all)
        break;
      check(tmx_sqp_run(ctx_, 1, nullptr));
    }
  }
  std::size_t numVars() const { return static_cast<std::size_t>(prob_->GetNumSteps()) * static_cast<std::size_t>(prob_->GetNumVarsPerStep()); }
  void check(tmx_status s)
  {
    if (s != TMX_OK)
      printAndThrow(std::string("libtrajopt_mi355x: ") + tmx_last_error(ctx_));
  }
  trajopt::TrajOptProb::Ptr prob_;
  tmx_ctx* ctx_{ nullptr };
  BasicTrustRegionSQPParameters param_;
  tmx_osqp_settings osqp_{};
  std::vector<double> seeds_;
  int32_t batch_{ 0 };
  OptResults results_;
  std::vector<OptResults> batch_results_;
  std::size_t best_{ 0 };
  std::vector<Callback> callbacks_;
  std::vector<IterationCallback> iteration_callbacks_;
  std::vector<ResultsCallback> results_callbacks_;
};

/** trajToDblVec (trajopt/include/trajopt/utils.hpp:18): row-major flattening of a TrajArray */
inline DblVec trajToDblVec(const TrajArray& t) { return t.data; }
}  // namespace sco

namespace trajopt
{
/** problem_description.hpp:112-121, problem_description.cpp:380-394 */
struct TrajOptResult
{
  using Ptr = std::shared_ptr<TrajOptResult>;
  std::vector<std::string> cost_names, cnt_names;
  DblVec cost_vals, cnt_viols;
  TrajArray traj;
  sco::OptStatus status;
  TrajOptResult(const sco::OptResults& opt, const TrajOptProb& prob)
    : cost_names(prob.getCostNames()), cnt_names(prob.getCntNames()), cost_vals(opt.cost_vals), cnt_viols(opt.cnt_viols), status(opt.status)
  {
    traj = TrajArray(prob.GetNumSteps(), prob.GetNumVarsPerStep());
    traj.data = opt.x;  // getTraj: the vars are laid out row-major (j_t_d)
  }
};

/** OptimizeProblem (problem_description.cpp:396-408): BasicTrustRegionSQP with the reference's planner-style parameters
    on the problem's own initial trajectory */
inline TrajOptResult::Ptr OptimizeProblem(const TrajOptProb::Ptr& prob, int device = 0)
{
  sco::BasicTrustRegionSQPBatchedHip opt(prob, device);
  sco::BasicTrustRegionSQPParameters& param = opt.getParameters();
  param.max_iter = 40;
  param.min_approx_improve_frac = .001;
  param.improve_ratio_threshold = .2;
  param.initial_merit_error_coeff = 20;
  opt.initialize(sco::trajToDblVec(prob->GetInitTraj()));
  opt.optimize();
  return std::make_shared<TrajOptResult>(opt.results(), *prob);
}
}  // namespace trajopt
}  // namespace tmx

#endif  // TMX_TRAJOPT_HPP_
