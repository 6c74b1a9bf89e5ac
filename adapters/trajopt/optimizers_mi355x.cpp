#include "optimizers_mi355x.hpp"

#include <trajopt_common/macros.h>
#include <tesseract/environment/environment.h>
#include <tesseract/kinematics/joint_group.h>
#include <tesseract/scene_graph/graph.h>
#include <tesseract/scene_graph/joint.h>
#include <tesseract/scene_graph/link.h>
#include <tesseract/geometry/impl/sphere.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <sstream>

namespace trajopt
{
namespace
{
void toRowMajor34(const Eigen::Isometry3d& T, double out[12])
{
  for (int r = 0; r < 3; ++r)
  {
    for (int c = 0; c < 3; ++c)
      out[4 * r + c] = T.linear()(r, c);
    out[4 * r + 3] = T.translation()(r);
  }
}
tmx_term blankTerm()
{
  tmx_term t;
  std::memset(&t, 0, sizeof(t));
  return t;
}
bool allZero(const DblVec& v) { return std::all_of(v.begin(), v.end(), [](double x) { return std::abs(x) < 1e-5; }); }
void fill(double* dst, const DblVec& src, std::size_t n, const char* what)
{
  if (src.size() != n)
    PRINT_AND_THROW(std::string("wrong number of values in ") + what);
  std::copy(src.begin(), src.end(), dst);
}

// CartPoseTermInfo / DynamicCartPoseTermInfo tolerance bands (validateTolerances, kinematic_terms.cpp:41-55) -> tmx_term::lower_tols /
// upper_tols [0..5]
void fillPoseTolerances(tmx_term& t, const Eigen::VectorXd& lower, const Eigen::VectorXd& upper, const std::string& name)
{
  if (lower.size() != upper.size())
    PRINT_AND_THROW(name + ": Mismatched tolerance sizes.");
  if (lower.size() == 0)
    return;
  if (lower.size() != 6)
    PRINT_AND_THROW(name + ": pose tolerances have six values (the rows of calcTransformError)");
  for (Eigen::Index i = 0; i < 6; ++i)
  {
    if (lower(i) > upper(i))
      PRINT_AND_THROW(name + ": Inverted tolerance band - lower > upper at one or more indices.");
    t.lower_tols[i] = lower(i);
    t.upper_tols[i] = upper(i);
  }
}

// index k of the moving link `name` = child link of joint k of the manipulator (-1: not one)
int movingLinkIndex(const ProblemConstructionInfo& pci, const std::string& name)
{
  const auto graph = pci.env->getSceneGraph();
  const std::vector<std::string> joint_names = pci.kin->getJointNames();
  for (std::size_t k = 0; k < joint_names.size(); ++k)
    if (graph->getJoint(joint_names[k])->child_link_name == name)
      return static_cast<int>(k);
  return -1;
}

// serial chain of the manipulator from the scene graph: per active joint the fixed transform from the previous moving link
// to the joint frame (folding the fixed joints in between), its axis and type
void lowerKinematics(const ProblemConstructionInfo& pci, LoweredProblem& out)
{
  const auto& kin = *pci.kin;
  const auto graph = pci.env->getSceneGraph();
  const std::vector<std::string> joint_names = kin.getJointNames();
  const auto n_dof = static_cast<int>(kin.numJoints());
  if (n_dof > TMX_MAX_DOF)
    PRINT_AND_THROW("manipulator has more joints than TMX_MAX_DOF");
  tmx_problem_desc& d = out.desc;
  d.n_dof = n_dof;
  const Eigen::MatrixX2d limits = kin.getLimits().joint_limits;
  const Eigen::VectorXd zeros = Eigen::VectorXd::Zero(n_dof);
  const tesseract::common::TransformMap tf0 = pci.env->getState(joint_names, zeros).link_transforms;
  Eigen::Isometry3d prev_link = Eigen::Isometry3d::Identity();  // world
  for (int k = 0; k < n_dof; ++k)
  {
    const auto joint = graph->getJoint(joint_names[static_cast<std::size_t>(k)]);
    tmx_joint& j = d.joints[k];
    switch (joint->type)
    {
      case tesseract::scene_graph::JointType::REVOLUTE:
      case tesseract::scene_graph::JointType::CONTINUOUS:
        j.type = 0;
        break;
      case tesseract::scene_graph::JointType::PRISMATIC:
        j.type = 1;
        break;
      default:
        PRINT_AND_THROW("joint type of " + joint->getName() + " is not lowered by the device path");
    }
    // at q = 0 the child link frame coincides with the joint frame
    const Eigen::Isometry3d joint_frame = tf0.at(joint->parent_link_name) * joint->parent_to_joint_origin_transform;
    toRowMajor34(prev_link.inverse() * joint_frame, j.origin);
    for (int a = 0; a < 3; ++a)
      j.axis[a] = joint->axis(a);
    d.joint_lower[k] = limits(k, 0);
    d.joint_upper[k] = limits(k, 1);
    prev_link = tf0.at(joint->child_link_name);
    // link spheres: sphere collision geometry of every link rigidly attached to this moving link
    for (const auto& link_name : kin.getActiveLinkNames())
    {
      // the closest moving ancestor of link_name is joint k's child iff no later active joint lies on its path to the root
      const auto link = graph->getLink(link_name);
      std::string cur = link_name;
      int owner = -1;
      while (owner < 0)
      {
        const auto in = graph->getInboundJoints(cur);
        if (in.empty())
          break;
        const auto it = std::find(joint_names.begin(), joint_names.end(), in.front()->getName());
        if (it != joint_names.end())
          owner = static_cast<int>(it - joint_names.begin());
        else
          cur = in.front()->parent_link_name;
      }
      if (owner != k)
        continue;
      for (const auto& col : link->collision)
      {
        // spheres, and capsules (tesseract: a cylinder of `length` along the local z axis with hemispherical caps) as the sphere
        // swept from one cap centre to the other (tmx_problem_desc::link_sphere_axes; discrete evaluators only)
        const auto type = col->geometry->getType();
        if (type != tesseract::geometry::GeometryType::SPHERE && type != tesseract::geometry::GeometryType::CAPSULE &&
            type != tesseract::geometry::GeometryType::BOX)
          // (convex meshes - tesseract::geometry::ConvexMesh, e.g. URDFs with tesseract:make_convex - go the same way as boxes: their
          //  vertices as a hull; the class is not part of the trajopt tree this adapter is type-checked against)
          PRINT_AND_THROW("collision geometry of link " + link_name + " is not a sphere, capsule or box: not lowered by the device path");
        const Eigen::Isometry3d g = prev_link.inverse() * tf0.at(link_name) * col->origin;
        if (type == tesseract::geometry::GeometryType::BOX)
        {
          // a box LINK is a convex hull of its eight corners in the moving link's frame (tmx_problem_desc::link_hull; contacts by
          // GJK / EPA, include/tmx_gjk.h) - boxbot.urdf of trajopt/test/cast_cost_unit.cpp
          const auto box = std::static_pointer_cast<const tesseract::geometry::Box>(col->geometry);
          tmx_link_sphere hs{};
          hs.link = k;
          hs.radius = 0.0;
          while (out.link_hull.size() < 2 * out.link_spheres.size())
            out.link_hull.push_back(0);
          out.link_hull.push_back(static_cast<int32_t>(out.hull_vertices.size() / 3));
          out.link_hull.push_back(8);
          for (int sx = -1; sx <= 1; sx += 2)
            for (int sy = -1; sy <= 1; sy += 2)
              for (int sz = -1; sz <= 1; sz += 2)
              {
                const Eigen::Vector3d v = g * Eigen::Vector3d(0.5 * sx * box->getX(), 0.5 * sy * box->getY(), 0.5 * sz * box->getZ());
                for (int q = 0; q < 3; ++q)
                  out.hull_vertices.push_back(v[q]);
              }
          out.link_spheres.push_back(hs);
          for (int q = 0; q < 3; ++q)
            out.link_sphere_axes.push_back(0.0);
          continue;
        }
        double radius = 0.0, half = 0.0;
        if (type == tesseract::geometry::GeometryType::SPHERE)
          radius = std::static_pointer_cast<const tesseract::geometry::Sphere>(col->geometry)->getRadius();
        else
        {
          const auto cap = std::static_pointer_cast<const tesseract::geometry::Capsule>(col->geometry);
          radius = cap->getRadius();
          half = 0.5 * cap->getLength();
        }
        const Eigen::Vector3d c = g * Eigen::Vector3d(0, 0, -half);
        const Eigen::Vector3d a = g * Eigen::Vector3d(0, 0, half) - c;  // cap centre to cap centre
        tmx_link_sphere s{};
        s.link = k;
        s.center[0] = c.x();
        s.center[1] = c.y();
        s.center[2] = c.z();
        s.radius = radius;
        out.link_spheres.push_back(s);
        for (int q = 0; q < 3; ++q)
          out.link_sphere_axes.push_back(a[q]);
      }
    }
  }
  toRowMajor34(Eigen::Isometry3d::Identity(), d.base);
  toRowMajor34(Eigen::Isometry3d::Identity(), d.tool);
  // static obstacles: sphere collision geometry of the links the manipulator does not move
  const std::vector<std::string> active = kin.getActiveLinkNames();
  for (const auto& link : graph->getLinks())
  {
    if (std::find(active.begin(), active.end(), link->getName()) != active.end())
      continue;
    for (const auto& col : link->collision)
    {
      const auto type = col->geometry->getType();
      if (type != tesseract::geometry::GeometryType::SPHERE && type != tesseract::geometry::GeometryType::CAPSULE &&
          type != tesseract::geometry::GeometryType::BOX)
        // (the C-ABI also takes convex triangle meshes - tmx_problem_desc::obstacle_mesh - for tesseract::geometry::ConvexMesh; that
        //  class is not part of the trajopt tree this adapter is type-checked against, so its flattening is left to the integrator)
        PRINT_AND_THROW("collision geometry of link " + link->getName() + " is not a sphere, capsule or box: not lowered by the device path");
      const Eigen::Isometry3d g = tf0.at(link->getName()) * col->origin;
      double radius = 0.0, half = 0.0;
      if (type == tesseract::geometry::GeometryType::BOX)
      {
        // tesseract::geometry::Box: full side lengths x, y, z about the geometry origin -> tmx_problem_desc::obstacle_boxes
        const auto box = std::static_pointer_cast<const tesseract::geometry::Box>(col->geometry);
        const Eigen::Vector3d c0 = g.translation();
        tmx_obstacle_sphere o{};
        o.center[0] = c0.x();
        o.center[1] = c0.y();
        o.center[2] = c0.z();
        o.radius = 0.0;
        out.obstacles.push_back(o);
        for (int q = 0; q < 3; ++q)
          out.obstacle_axes.push_back(0.0);
        out.obstacle_boxes.push_back(0.5 * box->getX());
        out.obstacle_boxes.push_back(0.5 * box->getY());
        out.obstacle_boxes.push_back(0.5 * box->getZ());
        for (int r = 0; r < 3; ++r)  // world_R_box row-major: row r = (ex[r], ey[r], ez[r]), the images of the box axes
        {
          const Eigen::Vector3d ex = g * Eigen::Vector3d(1, 0, 0) - c0, ey = g * Eigen::Vector3d(0, 1, 0) - c0, ez = g * Eigen::Vector3d(0, 0, 1) - c0;
          out.obstacle_boxes.push_back(ex[r]);
          out.obstacle_boxes.push_back(ey[r]);
          out.obstacle_boxes.push_back(ez[r]);
        }
        continue;
      }
      for (int q = 0; q < 12; ++q)
        out.obstacle_boxes.push_back(0.0);
      if (type == tesseract::geometry::GeometryType::SPHERE)
        radius = std::static_pointer_cast<const tesseract::geometry::Sphere>(col->geometry)->getRadius();
      else
      {
        const auto cap = std::static_pointer_cast<const tesseract::geometry::Capsule>(col->geometry);
        radius = cap->getRadius();
        half = 0.5 * cap->getLength();
      }
      const Eigen::Vector3d c = g * Eigen::Vector3d(0, 0, -half);
      const Eigen::Vector3d a = g * Eigen::Vector3d(0, 0, half) - c;  // cap centre to cap centre
      tmx_obstacle_sphere o{};
      o.center[0] = c.x();
      o.center[1] = c.y();
      o.center[2] = c.z();
      o.radius = radius;
      out.obstacles.push_back(o);
      for (int q = 0; q < 3; ++q)
        out.obstacle_axes.push_back(a[q]);
    }
  }
}

void lowerTerm(const ProblemConstructionInfo& pci, const TermInfo& ti, bool is_cost, const TrajArray& init_traj, int max_substates,
               LoweredProblem& out)
{
  const int n_steps = pci.basic_info.n_steps;
  const auto D = static_cast<std::size_t>(pci.kin->numJoints());
  auto clampSteps = [n_steps](int& first, int& last) {  // JointVelTermInfo::hatch, problem_description.cpp:1212-1226
    if ((n_steps - 2) <= first)
      first = n_steps - 2;
    if ((n_steps - 1) <= last)
      last = n_steps - 1;
    if (last == first)
      last += 1;
    if (last < first)
      std::swap(first, last);
  };
  tmx_term t = blankTerm();
  t.is_constraint = is_cost ? 0 : 1;
  std::vector<int32_t> fixed;
  std::vector<std::string> names{ ti.name };
  const bool with_time = static_cast<bool>(ti.term_type & TermType::TT_USE_TIME);
  if (with_time && !pci.basic_info.use_time)  // ConstructProblem, problem_description.cpp:447-448
    PRINT_AND_THROW("A term is using time and basic_info is not set correctly. Try basic_info.use_time = true");
  if (with_time && !static_cast<bool>(ti.getSupportedTypes() & TermType::TT_USE_TIME))  // :427-428, :441-442
    PRINT_AND_THROW(ti.name + " does not support time, but you listed it as a using time");
  if (const auto* tt = dynamic_cast<const TotalTimeTermInfo*>(&ti))
  {
    // TotalTimeTermInfo::hatch  problem_description.cpp:1852-1890
    if (!pci.basic_info.use_time)
      PRINT_AND_THROW(ti.name + ": TotalTime needs the time variables (basic_info.use_time)");
    t.kind = TMX_TERM_TOTAL_TIME;
    t.first_step = 1;
    t.last_step = n_steps - 1;
    t.coeff = tt->coeff;
    t.margin = tt->limit;
  }
  else if (const auto* jp = dynamic_cast<const JointPosTermInfo*>(&ti))
  {
    DblVec up = jp->upper_tols.empty() ? DblVec(D, 0) : jp->upper_tols, lo = jp->lower_tols.empty() ? DblVec(D, 0) : jp->lower_tols;
    const bool zero = allZero(up) && allZero(lo);  // :1113-1116
    t.kind = is_cost ? (zero ? TMX_TERM_JOINT_POS_EQ_COST : TMX_TERM_JOINT_POS_INEQ_COST) :
                       (zero ? TMX_TERM_JOINT_POS_EQ_CNT : TMX_TERM_JOINT_POS_INEQ_CNT);
    t.first_step = jp->first_step;
    t.last_step = jp->last_step;
    fill(t.coeffs, jp->coeffs.size() == 1 ? DblVec(D, jp->coeffs[0]) : jp->coeffs, D, "coeffs");
    fill(t.targets, jp->targets, D, "targets");
    fill(t.upper_tols, up, D, "upper_tols");
    fill(t.lower_tols, lo, D, "lower_tols");
  }
  else if (const auto* jv = dynamic_cast<const JointVelTermInfo*>(&ti))
  {
    DblVec up = jv->upper_tols.empty() ? DblVec(D, 0) : jv->upper_tols, lo = jv->lower_tols.empty() ? DblVec(D, 0) : jv->lower_tols;
    const bool zero = allZero(up) && allZero(lo);
    t.kind = is_cost ? (zero ? TMX_TERM_JOINT_VEL_COST : TMX_TERM_JOINT_VEL_INEQ_COST) :
                       (zero ? TMX_TERM_JOINT_VEL_EQ_CNT : TMX_TERM_JOINT_VEL_INEQ_CNT);
    if (with_time)
    {
      // :1244-1325: one TrajOptCostFromErrFunc / TrajOptConstraintFromErrFunc per joint, "name_j<j>"; expanded by the library
      t.kind = TMX_TERM_JOINT_VEL_TIME;
      names.clear();
      for (std::size_t j = 0; j < D; ++j)
        names.push_back(ti.name + "_j" + std::to_string(j));
    }
    int first = jv->first_step, last = jv->last_step;
    clampSteps(first, last);
    t.first_step = first;
    t.last_step = last;
    fill(t.coeffs, jv->coeffs.size() == 1 ? DblVec(D, jv->coeffs[0]) : jv->coeffs, D, "coeffs");
    fill(t.targets, jv->targets, D, "targets");
    fill(t.upper_tols, up, D, "upper_tols");
    fill(t.lower_tols, lo, D, "lower_tols");
  }
  else if (dynamic_cast<const JointAccTermInfo*>(&ti) != nullptr || dynamic_cast<const JointJerkTermInfo*>(&ti) != nullptr)
  {
    // JointAccTermInfo::hatch / JointJerkTermInfo::hatch (problem_description.cpp:1393-1493, :1515-1615): the same fields as
    // JointVelTermInfo, second / third difference; the library solves these problems with its dense QP engine
    const auto* ja = dynamic_cast<const JointAccTermInfo*>(&ti);
    const auto* jj = dynamic_cast<const JointJerkTermInfo*>(&ti);
    const int ord = ja != nullptr ? 2 : 3;
    const DblVec& c_in = ja ? ja->coeffs : jj->coeffs;
    const DblVec& t_in = ja ? ja->targets : jj->targets;
    const DblVec& u_in = ja ? ja->upper_tols : jj->upper_tols;
    const DblVec& l_in = ja ? ja->lower_tols : jj->lower_tols;
    DblVec up = u_in.empty() ? DblVec(D, 0) : u_in, lo = l_in.empty() ? DblVec(D, 0) : l_in;
    const bool zero = allZero(up) && allZero(lo);
    if (ord == 2)
      t.kind = is_cost ? (zero ? TMX_TERM_JOINT_ACC_EQ_COST : TMX_TERM_JOINT_ACC_INEQ_COST) :
                         (zero ? TMX_TERM_JOINT_ACC_EQ_CNT : TMX_TERM_JOINT_ACC_INEQ_CNT);
    else
      t.kind = is_cost ? (zero ? TMX_TERM_JOINT_JERK_EQ_COST : TMX_TERM_JOINT_JERK_INEQ_COST) :
                         (zero ? TMX_TERM_JOINT_JERK_EQ_CNT : TMX_TERM_JOINT_JERK_INEQ_CNT);
    int first = ja ? ja->first_step : jj->first_step, last = ja ? ja->last_step : jj->last_step;
    if (last <= -1)
      last = n_steps - 1;
    if ((n_steps - 1 - ord) <= first)  // :1407-1421 / :1529-1543
      first = n_steps - 1 - ord;
    if ((n_steps - 1) <= last)
      last = n_steps - 1;
    if (last == first)
      last += (ord == 2 ? 2 : 4);
    if (last < first)
      std::swap(first, last);
    t.first_step = first;
    t.last_step = last;
    fill(t.coeffs, c_in.empty() ? DblVec(D, 1) : (c_in.size() == 1 ? DblVec(D, c_in[0]) : c_in), D, "coeffs");
    fill(t.targets, t_in, D, "targets");
    fill(t.upper_tols, up, D, "upper_tols");
    fill(t.lower_tols, lo, D, "lower_tols");
  }
  else if (const auto* cp = dynamic_cast<const CartPoseTermInfo*>(&ti))
  {
    if (cp->error_function != nullptr)
      PRINT_AND_THROW(ti.name + ": custom CartPose error functions are not lowered by the device path");
    fillPoseTolerances(t, cp->lower_tolerance, cp->upper_tolerance, ti.name);
    const auto tip = pci.kin->getActiveLinkNames().back();
    if (cp->source_frame != tip)
      PRINT_AND_THROW(ti.name + ": the source frame must be the manipulator's tip link (one tool frame per problem)");
    if (pci.kin->isActiveLinkName(cp->target_frame))
      PRINT_AND_THROW(ti.name + ": a moving target frame is the reference's DynamicCartPose term: not lowered");
    t.kind = TMX_TERM_CART_POSE;
    t.first_step = t.last_step = cp->timestep;
    for (int i = 0; i < 3; ++i)
    {
      t.coeffs[i] = cp->pos_coeffs(i);
      t.coeffs[3 + i] = cp->rot_coeffs(i);
    }
    toRowMajor34(pci.env->getLinkTransform(cp->target_frame) * cp->target_frame_offset, t.target_pose);
    out.setTool(cp->source_frame_offset, ti.name);  // tcp offset on the tip link
  }
  else if (const auto* col = dynamic_cast<const CollisionTermInfo*>(&ti))
  {
    if (!col->config.enabled)
      return;
    const auto type = col->config.collision_check_config.type;
    t.kind = is_cost ? TMX_TERM_COLLISION_COST : TMX_TERM_COLLISION_CNT;
    t.first_step = col->first_step;
    t.last_step = col->last_step;
    t.margin = col->config.contact_manager_config.default_margin.value_or(0.0);
    t.coeff = col->config.collision_coeff_data.getCollisionCoeff("", "");  // default coefficient; per-pair overrides throw below
    if (!col->config.collision_coeff_data.getPairsWithZeroCoeff().empty())
      PRINT_AND_THROW(ti.name + ": per-pair collision coefficients are not lowered by the device path");
    t.buffer = col->config.collision_margin_buffer;
    t.evaluator_type = static_cast<int32_t>(type);
    t.longest_valid_segment_length = col->config.collision_check_config.longest_valid_segment_length;
    t.max_substates = max_substates;
    if (t.evaluator_type >= 2 && t.max_substates <= 0)
    {
      double dmax = 0.0;
      for (Eigen::Index i = 0; i + 1 < init_traj.rows(); ++i)
        dmax = std::max(dmax, (init_traj.row(i + 1) - init_traj.row(i)).norm());
      t.max_substates = static_cast<int32_t>(
          std::min(64.0, std::max(2.0, std::ceil(1.5 * dmax / std::max(t.longest_valid_segment_length, 1e-9)) + 1.0)));
    }
    fixed.assign(col->fixed_steps.begin(), col->fixed_steps.end());
    names.clear();
    if (t.evaluator_type >= 2)
      for (int i = t.first_step; i < t.last_step; ++i)
        names.push_back(ti.name + "_" + std::to_string(i));
    else
      for (int i = t.first_step; i <= t.last_step; ++i)
        if (std::find(fixed.begin(), fixed.end(), i) == fixed.end())
          names.push_back(ti.name + "_" + std::to_string(i));
  }
  else if (const auto* dcp = dynamic_cast<const DynamicCartPoseTermInfo*>(&ti))
  {
    // DynamicCartPoseTermInfo::hatch (problem_description.cpp:752-822): source = the tool frame, target = a moving link of the chain
    fillPoseTolerances(t, dcp->lower_tolerance, dcp->upper_tolerance, ti.name);
    if (static_cast<bool>(ti.term_type & TermType::TT_USE_TIME))
      PRINT_AND_THROW(ti.name + ": Use time version of this term has not been defined.");
    if (dcp->source_frame != pci.kin->getActiveLinkNames().back())
      PRINT_AND_THROW(ti.name + ": the source frame must be the manipulator's tip link (one tool frame per problem)");
    const int target = movingLinkIndex(pci, dcp->target_frame);
    if (target < 0)
      PRINT_AND_THROW(ti.name + ": the target frame must be the child link of one of the manipulator's joints");
    t.kind = TMX_TERM_DYN_CART_POSE;
    t.first_step = t.last_step = dcp->timestep;
    for (int i = 0; i < 3; ++i)
    {
      t.coeffs[i] = dcp->pos_coeffs(i);
      t.coeffs[3 + i] = dcp->rot_coeffs(i);
    }
    t.link = target;
    toRowMajor34(dcp->target_frame_offset, t.target_pose);
    out.setTool(dcp->source_frame_offset, ti.name);  // tcp offset on the tip link
  }
  else if (const auto* avs = dynamic_cast<const AvoidSingularityTermInfo*>(&ti))
  {
    // AvoidSingularityTermInfo::hatch (problem_description.cpp:1900-1940), the problem's full joint set; names name_<step>
    const int link = movingLinkIndex(pci, avs->link);
    if (link < 0)
      PRINT_AND_THROW(ti.name + ": the link must be the child link of one of the manipulator's joints");
    if (avs->subset_kin_ != nullptr)
    {
      // the subset form is used when the subset's joints are joints of the problem (isSuperset, :1907); lowered when they are the run
      // of joints first .. link of the chain
      const std::vector<std::string> all = pci.kin->getJointNames(), sub = avs->subset_kin_->getJointNames();
      bool subset = !sub.empty();
      for (const auto& n : sub)
        subset = subset && std::find(all.begin(), all.end(), n) != all.end();
      if (subset)
      {
        const auto first = std::find(all.begin(), all.end(), sub.front());
        const int j0 = static_cast<int>(first - all.begin());
        if (j0 + static_cast<int>(sub.size()) != link + 1 || !std::equal(sub.begin(), sub.end(), first))
          PRINT_AND_THROW(ti.name + ": the joint subset must be a run of the manipulator's joints that ends at the link's joint");
        t.subset_first = j0 + 1;
      }
    }
    if (avs->coeffs.size() != 1)
      PRINT_AND_THROW(ti.name + ": one coefficient (the error has one row)");
    t.kind = TMX_TERM_AVOID_SINGULARITY;
    t.first_step = avs->first_step;
    t.last_step = avs->last_step;
    t.coeffs[0] = avs->coeffs[0];
    t.link = link;
    t.lambda = avs->lambda;
    names.clear();
    for (int i = t.first_step; i <= t.last_step; ++i)
      names.push_back(ti.name + "_" + std::to_string(i));
  }
  else if (const auto* cv = dynamic_cast<const CartVelTermInfo*>(&ti))
  {
    // CartVelTermInfo::hatch (problem_description.cpp:1011-1057): one cost (named after the term) / one constraint "CartVel"
    // per step i in [first_step, last_step] over waypoints i and i + 1
    if (static_cast<bool>(ti.term_type & TermType::TT_USE_TIME))
      PRINT_AND_THROW(ti.name + ": Use time version of this term has not been defined.");
    if (cv->link != pci.kin->getActiveLinkNames().back())
      PRINT_AND_THROW(ti.name + ": the link must be the manipulator's tip link (one tool frame per problem)");
    t.kind = TMX_TERM_CART_VEL;
    t.first_step = cv->first_step;
    t.last_step = cv->last_step;
    t.margin = cv->max_displacement;
    names.assign(static_cast<std::size_t>(cv->last_step - cv->first_step + 1), is_cost ? ti.name : std::string("CartVel"));
  }
  else
    PRINT_AND_THROW("term \"" + ti.name + "\" has a TermInfo class the device path does not lower (UserDefinedTermInfo with "
                    "opaque callbacks): solve it with the reference's "
                    "BasicTrustRegionSQP, or with its QPs on the device through HipBatchedAdmmModel");
  auto& dst = is_cost ? out.cost_names : out.cnt_names;
  dst.insert(dst.end(), names.begin(), names.end());
  out.terms.push_back(t);
  out.term_fixed_steps.push_back(fixed);
}
}  // namespace

// The reference keeps source_frame_offset per term (CartPoseTermInfo / DynamicCartPoseTermInfo); the device keeps one tool offset
// per problem.  Two pose terms with different offsets must not silently share the last writer's: explicit error.
void LoweredProblem::setTool(const Eigen::Isometry3d& source_frame_offset, const std::string& term_name)
{
  double m[12];
  toRowMajor34(source_frame_offset, m);
  if (have_tool)
  {
    for (int q = 0; q < 12; ++q)
      if (m[q] != desc.tool[q])
        PRINT_AND_THROW(term_name + ": pose terms with different source_frame_offset in one problem are not lowered by the device path "
                                    "(one tool frame per problem)");
    return;
  }
  for (int q = 0; q < 12; ++q)
    desc.tool[q] = m[q];
  have_tool = true;
}

void LoweredProblem::finalize()
{
  for (std::size_t k = 0; k < terms.size(); ++k)
  {
    terms[k].n_fixed_steps = static_cast<int32_t>(term_fixed_steps[k].size());
    terms[k].fixed_steps = term_fixed_steps[k].empty() ? nullptr : term_fixed_steps[k].data();
  }
  desc.n_link_spheres = static_cast<int32_t>(link_spheres.size());
  desc.link_spheres = link_spheres.data();
  if (!hull_vertices.empty())
  {
    link_hull.resize(2 * link_spheres.size(), 0);
    desc.link_hull = link_hull.data();
    desc.hull_vertices = hull_vertices.data();
    desc.n_hull_vertices = static_cast<int32_t>(hull_vertices.size() / 3);
  }
  desc.link_sphere_axes = (link_sphere_axes.size() == 3 * link_spheres.size() && !link_spheres.empty()) ? link_sphere_axes.data() : nullptr;
  desc.obstacle_axes = (obstacle_axes.size() == 3 * obstacles.size() && !obstacles.empty()) ? obstacle_axes.data() : nullptr;
  desc.obstacle_boxes = (obstacle_boxes.size() == 12 * obstacles.size() && !obstacles.empty()) ? obstacle_boxes.data() : nullptr;
  desc.n_obstacles = static_cast<int32_t>(obstacles.size());
  desc.obstacles = obstacles.data();
  desc.n_fixed_steps = static_cast<int32_t>(fixed_steps.size());
  desc.fixed_steps = fixed_steps.data();
  desc.n_fixed_dofs = static_cast<int32_t>(fixed_dofs.size());
  desc.fixed_dofs = fixed_dofs.data();
  desc.n_terms = static_cast<int32_t>(terms.size());
  desc.terms = terms.data();
}

LoweredProblem lowerProblem(const ProblemConstructionInfo& pci, const TrajArray& init_traj, int max_substates)
{
  LoweredProblem out;
  out.desc.n_steps = pci.basic_info.n_steps;
  // time-parameterised problems: the caller's init_traj carries the time column (generateInitTraj, problem_description.cpp:367-376)
  out.desc.use_time = pci.basic_info.use_time ? 1 : 0;
  out.desc.dt_lower_lim = pci.basic_info.dt_lower_lim;
  out.desc.dt_upper_lim = pci.basic_info.dt_upper_lim;
  if (pci.basic_info.use_time)
  {
    bool any = false;  // (:451-452)
    for (const auto& lst : { pci.cost_infos, pci.cnt_infos })
      for (const auto& ci : lst)
        any = any || static_cast<bool>(ci->term_type & TermType::TT_USE_TIME);
    if (!any)
      PRINT_AND_THROW("No terms use time and basic_info is not set correctly. Try basic_info.use_time = false");
  }
  lowerKinematics(pci, out);
  out.fixed_steps.assign(pci.basic_info.fixed_timesteps.begin(), pci.basic_info.fixed_timesteps.end());
  out.fixed_dofs.assign(pci.basic_info.fixed_dofs.begin(), pci.basic_info.fixed_dofs.end());
  for (const auto& ci : pci.cost_infos)
    lowerTerm(pci, *ci, true, init_traj, max_substates, out);
  for (const auto& ci : pci.cnt_infos)
    lowerTerm(pci, *ci, false, init_traj, max_substates, out);
  out.finalize();
  return out;
}
}  // namespace trajopt

namespace sco
{
namespace
{
tmx_sqp_params toParams(const BasicTrustRegionSQPParameters& p)
{
  tmx_sqp_params o;
  tmx_default_sqp_params(&o);
  o.improve_ratio_threshold = p.improve_ratio_threshold;
  o.min_trust_box_size = p.min_trust_box_size;
  o.min_approx_improve = p.min_approx_improve;
  o.min_approx_improve_frac = p.min_approx_improve_frac;
  o.max_iter = static_cast<int32_t>(p.max_iter);
  o.max_qp_solver_failures = static_cast<int32_t>(p.max_qp_solver_failures);
  o.trust_shrink_ratio = p.trust_shrink_ratio;
  o.trust_expand_ratio = p.trust_expand_ratio;
  o.cnt_tolerance = p.cnt_tolerance;
  o.max_merit_coeff_increases = p.max_merit_coeff_increases;
  o.merit_coeff_increase_ratio = p.merit_coeff_increase_ratio;
  o.initial_merit_error_coeff = p.initial_merit_error_coeff;
  o.inflate_constraints_individually = p.inflate_constraints_individually ? 1 : 0;
  o.trust_box_size = p.trust_box_size;
  return o;
}
}  // namespace

BasicTrustRegionSQPBatchedHip::BasicTrustRegionSQPBatchedHip(const std::shared_ptr<trajopt::TrajOptProb>& prob,
                                                             const trajopt::ProblemConstructionInfo& pci, int device)
  : BasicTrustRegionSQP(prob), lowered_(trajopt::lowerProblem(pci, prob->GetInitTraj()))
{
  if (tmx_create(device, &ctx_) != TMX_OK)
    PRINT_AND_THROW("BasicTrustRegionSQPBatchedHip: no usable HIP device (there is no CPU fallback behind this optimizer)");
}

BasicTrustRegionSQPBatchedHip::~BasicTrustRegionSQPBatchedHip() { tmx_destroy(ctx_); }

void BasicTrustRegionSQPBatchedHip::initializeBatch(const std::vector<DblVec>& seeds)
{
  const auto n = static_cast<std::size_t>(lowered_.desc.n_steps) * static_cast<std::size_t>(lowered_.desc.n_dof);
  for (const DblVec& s : seeds)
    if (s.size() != n)
      PRINT_AND_THROW("initialization vector has wrong length");  // Optimizer::initialize, optimizers.cpp:127-136
  seeds_ = seeds;
  if (!seeds_.empty())
    initialize(seeds_.front());
}

void BasicTrustRegionSQPBatchedHip::attachCommunicator(void* nccl_comm, long global_offset)
{
  global_offset_ = global_offset;
  if (tmx_attach_nccl(ctx_, nccl_comm) != TMX_OK)
    PRINT_AND_THROW(tmx_last_error(ctx_));
}

OptStatus BasicTrustRegionSQPBatchedHip::optimize()
{
  if (seeds_.empty())
    seeds_.push_back(results_.x);  // initialize(x) of the base class
  const auto B = static_cast<int32_t>(seeds_.size());
  const std::size_t n = seeds_.front().size();
  const tmx_sqp_params sp = toParams(param_);
  if (tmx_problem_upload(ctx_, &lowered_.desc, &sp, nullptr) != TMX_OK)
    PRINT_AND_THROW(tmx_last_error(ctx_));
  DblVec flat(static_cast<std::size_t>(B) * n);
  for (int32_t b = 0; b < B; ++b)
    std::copy(seeds_[static_cast<std::size_t>(b)].begin(), seeds_[static_cast<std::size_t>(b)].end(), flat.begin() + b * static_cast<long>(n));
  if (tmx_batch_set_x0(ctx_, flat.data(), B) != TMX_OK || tmx_sqp_run(ctx_, 0, nullptr) != TMX_OK)
    PRINT_AND_THROW(tmx_last_error(ctx_));
  std::vector<int32_t> status(static_cast<std::size_t>(B)), nfe(status.size()), nqp(status.size());
  DblVec cost(status.size());
  int32_t n_costs = 0, n_cnts = 0;
  tmx_term_counts(ctx_, &n_costs, &n_cnts, nullptr);
  DblVec cost_vals(status.size() * static_cast<std::size_t>(n_costs)), cnt_viols(status.size() * static_cast<std::size_t>(n_cnts));
  if (tmx_sqp_results(ctx_, flat.data(), status.data(), cost.data(), nfe.data(), nqp.data()) != TMX_OK ||
      tmx_evaluate(ctx_, cost_vals.data(), cnt_viols.data()) != TMX_OK)
    PRINT_AND_THROW(tmx_last_error(ctx_));
  batch_results_.assign(status.size(), OptResults());
  for (std::size_t b = 0; b < status.size(); ++b)
  {
    OptResults& r = batch_results_[b];
    r.x.assign(flat.begin() + static_cast<long>(b * n), flat.begin() + static_cast<long>((b + 1) * n));
    r.status = static_cast<OptStatus>(status[b]);
    r.total_cost = cost[b];
    r.cost_vals.assign(cost_vals.begin() + static_cast<long>(b) * n_costs, cost_vals.begin() + static_cast<long>(b + 1) * n_costs);
    r.cnt_viols.assign(cnt_viols.begin() + static_cast<long>(b) * n_cnts, cnt_viols.begin() + static_cast<long>(b + 1) * n_cnts);
    r.n_func_evals = nfe[b];
    r.n_qp_solves = nqp[b];
  }
  int64_t best = -1;
  double best_cost = 0;
  if (tmx_argmin(ctx_, global_offset_, &best, &best_cost) != TMX_OK)
    PRINT_AND_THROW(tmx_last_error(ctx_));
  best_ = (best >= global_offset_ && best < global_offset_ + B) ? static_cast<long>(best - global_offset_) : -1;
  results_ = batch_results_[static_cast<std::size_t>(best_ >= 0 ? best_ : 0)];
  callCallbacks();  // "at exit" call of the reference (optimizers.cpp:978) on the reported seed
  return results_.status;
}
}  // namespace sco
