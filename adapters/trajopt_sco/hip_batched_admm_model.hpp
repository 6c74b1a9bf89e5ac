/**
 * HipBatchedAdmmModel — sco::Model backed by libtrajopt_mi355x.so (tmx_qp_solve_batched).
 * Surface: trajopt_sco/include/trajopt_sco/solver_interface.hpp:54-104; behaviour of the bookkeeping follows
 * OSQPModel (trajopt_sco/src/osqp_interface.cpp:123-168, 372-438, 440-615).
 * Compiled inside a trajopt checkout (needs trajopt_sco + Eigen); see adapters/README.md.
 */
#pragma once
#include <trajopt_sco/solver_interface.hpp>
#include <trajopt_sco/osqp_interface.hpp>  // OSQPModelConfig: the settings callers already hand to trajopt

#include <mutex>
#include <tmx.h>

namespace sco
{
/** selects the MI355X back-end through createModel(ModelType::OSQP, config) (see solver_interface_mi355x.patch) */
struct HipModelConfig : public ModelConfig
{
  using Ptr = std::shared_ptr<HipModelConfig>;
  using ConstPtr = std::shared_ptr<const HipModelConfig>;
  int device{ 0 };
  tmx_osqp_settings settings;
  HipModelConfig() { tmx_default_osqp_settings(&settings); }
  /** same fields the reference sets on OSQPSettings (osqp_interface.cpp:78-90) */
  explicit HipModelConfig(const OSQPModelConfig& c);
};

class HipBatchedAdmmModel : public Model
{
public:
  explicit HipBatchedAdmmModel(const ModelConfig::ConstPtr& config = nullptr);
  ~HipBatchedAdmmModel() override;
  HipBatchedAdmmModel(const HipBatchedAdmmModel&) = delete;
  HipBatchedAdmmModel& operator=(const HipBatchedAdmmModel&) = delete;

  Var addVar(const std::string& name) override;
  Cnt addEqCnt(const AffExpr&, const std::string& name) override;
  Cnt addIneqCnt(const AffExpr&, const std::string& name) override;
  Cnt addIneqCnt(const QuadExpr&, const std::string& name) override;
  void removeVars(const VarVector& vars) override;
  void removeCnts(const CntVector& cnts) override;

  void update() override;
  void setVarBounds(const VarVector& vars, const DblVec& lower, const DblVec& upper) override;
  DblVec getVarValues(const VarVector& vars) const override;
  CvxOptStatus optimize() override;
  void setObjective(const AffExpr&) override;
  void setObjective(const QuadExpr&) override;
  void writeToFile(const std::string& fname) const override;
  VarVector getVars() const override;

private:
  tmx_ctx* ctx_{ nullptr };
  tmx_osqp_settings settings_;
  VarVector vars_;
  CntVector cnts_;
  DblVec lbs_, ubs_;
  AffExprVector cnt_exprs_;
  ConstraintTypeVector cnt_types_;
  QuadExpr objective_;
  DblVec solution_;
  // previous solve: the reference re-applies x, y and rho when the sparsity is unchanged (osqp_interface.cpp:338-369)
  std::vector<int64_t> prev_P_p_, prev_P_i_, prev_A_p_, prev_A_i_;
  std::size_t prev_n_{ 0 }, prev_m_{ 0 };  // dimensions of the previous QP (the reference compares them before the byte prefixes)
  DblVec prev_x_, prev_y_;
  double prev_rho_{ 0 };
  bool prev_solved_{ false };
  std::mutex mutex_;  // add / remove must be thread safe (solver_interface.hpp:67-91)
};
}  // namespace sco
