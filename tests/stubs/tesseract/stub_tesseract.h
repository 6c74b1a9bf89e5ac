// TEST SCAFFOLDING — declaration-level stand-in for the tesseract names that the reference's trajopt headers and the
// reference-side adapter adapters/trajopt/optimizers_mi355x.cpp mention (tesseract is an external dependency of the
// reference and is not in this container).  Member names follow tesseract 0.3x as the reference uses them
// (trajopt/src/problem_description.cpp, trajopt_common/src/collision_utils.cpp).  Used by tests/test_adapters_compile.py only.
#pragma once
#include <Eigen/Geometry>
#include <array>
#include <cstdint>
#include <map>
#include <memory>
#include <optional>
#include <set>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace tesseract
{
namespace common
{
using TransformMap = std::map<std::string, Eigen::Isometry3d>;
using LinkNamesPair = std::pair<std::string, std::string>;
using VectorIsometry3d = std::vector<Eigen::Isometry3d>;
template <typename T>
using AlignedVector = std::vector<T>;
using TrajArray = Eigen::MatrixXd;
struct CollisionMarginData  // tesseract/common/collision_margin_data.h: the two accessors the adapters read
{
  double getDefaultCollisionMargin() const;
  double getMaxCollisionMargin() const;
};
Eigen::VectorXd calcTransformError(const Eigen::Isometry3d& t1, const Eigen::Isometry3d& t2);
}  // namespace common
}  // namespace tesseract
namespace std
{
template <>
struct hash<tesseract::common::LinkNamesPair>  // tesseract/common/types.h provides this specialisation
{
  size_t operator()(const tesseract::common::LinkNamesPair& p) const;
};
}  // namespace std
namespace tesseract
{
namespace geometry
{
enum class GeometryType : std::uint8_t
{
  UNINITIALIZED,
  SPHERE,
  CYLINDER,
  CAPSULE,
  CONE,
  BOX,
  PLANE,
  MESH,
  CONVEX_MESH,
  SDF_MESH,
  OCTREE,
  POLYGON_MESH,
  COMPOUND_MESH
};
class Geometry
{
public:
  using ConstPtr = std::shared_ptr<const Geometry>;
  virtual ~Geometry() = default;
  GeometryType getType() const;
};
class Sphere : public Geometry
{
public:
  double getRadius() const;
};
class Capsule : public Geometry
{
public:
  double getRadius() const;
  double getLength() const;
};
class Box : public Geometry
{
public:
  double getX() const;
  double getY() const;
  double getZ() const;
};
}  // namespace geometry
namespace scene_graph
{
enum class JointType : std::uint8_t
{
  UNKNOWN,
  REVOLUTE,
  CONTINUOUS,
  PRISMATIC,
  FLOATING,
  PLANAR,
  FIXED
};
struct Collision
{
  using Ptr = std::shared_ptr<Collision>;
  std::string name;
  Eigen::Isometry3d origin;
  geometry::Geometry::ConstPtr geometry;
};
class Link
{
public:
  using ConstPtr = std::shared_ptr<const Link>;
  const std::string& getName() const;
  std::vector<Collision::Ptr> collision;
};
class Joint
{
public:
  using ConstPtr = std::shared_ptr<const Joint>;
  const std::string& getName() const;
  JointType type;
  Eigen::Vector3d axis;
  std::string child_link_name, parent_link_name;
  Eigen::Isometry3d parent_to_joint_origin_transform;
};
class SceneGraph
{
public:
  using ConstPtr = std::shared_ptr<const SceneGraph>;
  Joint::ConstPtr getJoint(const std::string& name) const;
  Link::ConstPtr getLink(const std::string& name) const;
  std::vector<Link::ConstPtr> getLinks() const;
  std::vector<Joint::ConstPtr> getInboundJoints(const std::string& link_name) const;
};
struct SceneState
{
  std::unordered_map<std::string, double> joints;
  common::TransformMap link_transforms;
  common::TransformMap joint_transforms;
};
}  // namespace scene_graph
namespace kinematics
{
struct KinematicLimits
{
  Eigen::MatrixX2d joint_limits;
};
class JointGroup
{
public:
  using ConstPtr = std::shared_ptr<const JointGroup>;
  std::vector<std::string> getJointNames() const;
  std::vector<std::string> getActiveLinkNames() const;
  std::vector<std::string> getLinkNames() const;
  std::string getBaseLinkName() const;
  bool isActiveLinkName(const std::string& link_name) const;
  Eigen::Index numJoints() const;
  KinematicLimits getLimits() const;
  common::TransformMap calcFwdKin(const Eigen::Ref<const Eigen::VectorXd>& joint_angles) const;
  Eigen::MatrixXd calcJacobian(const Eigen::Ref<const Eigen::VectorXd>& joint_angles, const std::string& link_name) const;
};
}  // namespace kinematics
namespace visualization
{
class Visualization;
}
namespace collision
{
class DiscreteContactManager;    // tesseract/collision/fwd.h
class ContinuousContactManager;
enum class ContinuousCollisionType : std::uint8_t
{
  CCType_None,
  CCType_Time0,
  CCType_Time1,
  CCType_Between
};
enum class CollisionEvaluatorType : std::uint8_t
{
  NONE,
  DISCRETE,
  LVS_DISCRETE,
  CONTINUOUS,
  LVS_CONTINUOUS
};
enum class CollisionCheckProgramType : std::uint8_t
{
  ALL,
  ALL_EXCEPT_START,
  ALL_EXCEPT_END,
  START_ONLY,
  END_ONLY,
  INTERMEDIATE_ONLY
};
struct ContactRequest
{
};
struct ContactManagerConfig
{
  std::optional<double> default_margin;
};
struct CollisionCheckConfig
{
  ContactRequest contact_request;
  CollisionEvaluatorType type{ CollisionEvaluatorType::DISCRETE };
  double longest_valid_segment_length{ 0.005 };
  CollisionCheckProgramType check_program_mode{ CollisionCheckProgramType::ALL };
};
struct ContactResult
{
  double distance{ 0 };
  std::array<std::string, 2> link_names;
  std::array<Eigen::Vector3d, 2> nearest_points, nearest_points_local;
  std::array<Eigen::Isometry3d, 2> transform, cc_transform;
  std::array<double, 2> cc_time;
  std::array<ContinuousCollisionType, 2> cc_type;
  Eigen::Vector3d normal;
};
using ContactResultVector = std::vector<ContactResult>;
class ContactResultMap
{
};
}  // namespace collision
namespace environment
{
class Environment
{
public:
  using Ptr = std::shared_ptr<Environment>;
  using ConstPtr = std::shared_ptr<const Environment>;
  scene_graph::SceneGraph::ConstPtr getSceneGraph() const;
  std::shared_ptr<const kinematics::JointGroup> getJointGroup(const std::string& name) const;
  scene_graph::SceneState getState() const;
  scene_graph::SceneState getState(const std::vector<std::string>& joint_names, const Eigen::Ref<const Eigen::VectorXd>& joint_values) const;
  std::vector<std::string> getLinkNames() const;
  std::vector<std::string> getActiveLinkNames() const;
  Eigen::Isometry3d getLinkTransform(const std::string& link_name) const;
};
}  // namespace environment
}  // namespace tesseract
