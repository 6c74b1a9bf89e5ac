/**
 * optimizers_mi355x — sco::BasicTrustRegionSQP for a BATCH of seeds of one trajopt problem on the MI355X
 * (S3 trajopt_sco/include/trajopt_sco/optimizers.hpp:137-218, S4 trajopt/include/trajopt/problem_description.hpp:199-667).
 *
 *   trajopt::ProblemConstructionInfo pci(env);  pci.fromJson(root);          // the reference's own JSON reader
 *   auto prob = trajopt::ConstructProblem(pci);                               // unchanged: callbacks, names, init traj
 *   sco::BasicTrustRegionSQPBatchedHip opt(prob, pci);                        // lowers the TermInfos (throws on unknown ones)
 *   opt.setParameters(pci.opt_info);
 *   opt.initializeBatch(seeds);                                               // B x (n_steps * n_dof); or initialize(x): B = 1
 *   opt.optimize();                                                           // whole SQP of every seed on the device
 *   opt.results();            // the best converged seed (argmin total_cost), as sco::OptResults
 *   opt.batchResults();       // every seed
 *
 * Compiled inside a trajopt checkout (needs trajopt, trajopt_sco, tesseract); see adapters/README.md.
 */
#pragma once
#include <trajopt_sco/optimizers.hpp>
#include <trajopt/problem_description.hpp>

#include <tmx.h>
#include <string>
#include <vector>

namespace trajopt
{
/** flat, self-contained copy of what the device path needs from a ProblemConstructionInfo: owns every array the
    tmx_problem_desc points to */
struct LoweredProblem
{
  tmx_problem_desc desc{};
  std::vector<tmx_term> terms;
  std::vector<std::vector<int32_t>> term_fixed_steps;
  std::vector<int32_t> fixed_steps, fixed_dofs;
  std::vector<tmx_link_sphere> link_spheres;
  std::vector<tmx_obstacle_sphere> obstacles;
  std::vector<double> link_sphere_axes, obstacle_axes;  // capsules: 3 per primitive (zero = sphere); empty when every primitive is a sphere
  std::vector<double> obstacle_boxes;                    // boxes: 12 per obstacle (half extents, rotation; zero = not a box)
  std::vector<int32_t> link_hull;                        // convex-hull links: 2 per link primitive (first vertex, count; 0 = sphere / capsule)
  std::vector<double> hull_vertices;                     // 3 per vertex, link frame
  std::vector<std::string> cost_names, cnt_names;
  bool have_tool{ false };  // a pose term has written desc.tool (the device keeps ONE tool offset per problem; the reference one per term)
  void setTool(const Eigen::Isometry3d& source_frame_offset, const std::string& term_name);
  void finalize();  // wires the pointers of `desc`
};

/** TermInfo::hatch for the device: throws std::runtime_error for every term class / option the device path does not lower
    (UserDefinedTermInfo with opaque callbacks, custom CartPose error functions, link geometry other than spheres and capsules, obstacle geometry other than spheres, capsules and boxes) - explicit, never a silent CPU detour.
    `max_substates`: row-slot capacity of the LVS / continuous collision evaluators (0 = from the initial trajectory). */
LoweredProblem lowerProblem(const ProblemConstructionInfo& pci, const TrajArray& init_traj, int max_substates = 0);
}  // namespace trajopt

namespace sco
{
class BasicTrustRegionSQPBatchedHip : public BasicTrustRegionSQP
{
public:
  using Ptr = std::shared_ptr<BasicTrustRegionSQPBatchedHip>;
  BasicTrustRegionSQPBatchedHip(const std::shared_ptr<trajopt::TrajOptProb>& prob, const trajopt::ProblemConstructionInfo& pci,
                                int device = 0);
  ~BasicTrustRegionSQPBatchedHip() override;
  BasicTrustRegionSQPBatchedHip(const BasicTrustRegionSQPBatchedHip&) = delete;
  BasicTrustRegionSQPBatchedHip& operator=(const BasicTrustRegionSQPBatchedHip&) = delete;

  /** Optimizer::initialize for B seeds (row-major B x n_vars); initialize(x) of the base class is the B = 1 case */
  void initializeBatch(const std::vector<DblVec>& seeds);
  OptStatus optimize() override;
  const std::vector<OptResults>& batchResults() const { return batch_results_; }
  /** index of the best converged seed in the last optimize() (-1: none converged) */
  long bestSeed() const { return best_; }
  /** attach an ncclComm_t: the best-seed reduction then runs over all ranks (tmx_attach_nccl) */
  void attachCommunicator(void* nccl_comm, long global_offset);

private:
  trajopt::LoweredProblem lowered_;
  tmx_ctx* ctx_{ nullptr };
  std::vector<DblVec> seeds_;
  std::vector<OptResults> batch_results_;
  long best_{ -1 };
  long global_offset_{ 0 };
};
}  // namespace sco
