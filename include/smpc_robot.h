/*
 * smpc_robot.h -- plain-C robot description table (kinematic tree + inertias + feet).
 *
 * Replaces, for the MPC hot path, what the reference obtains from a pinocchio::Model through
 * RobotModelHandler (reference: include/simple-mpc/robot-handler.hpp:28-225,
 * src/robot-handler.cpp:12-79).  Pinocchio and the example-robot-data URDFs are not available
 * in this build, so the robot is passed as a flat table.  Data only: no code.
 *
 * Conventions (identical to Pinocchio so that states are interchangeable):
 *   - joint 0 is the free-flyer: q[0:7] = [x y z qx qy qz qw], v[0:6] = [v_local; w_local]
 *   - joints 1..njoints-1 are revolute about a local axis (jtype 1/2/3 = X/Y/Z)
 *   - every joint carries one body; fixed children are already merged into it
 *   - inertia[] = {Ixx, Ixy, Iyy, Ixz, Iyz, Izz} about the body CoM, in the joint frame
 *   - a foot is an operational frame with identity rotation at foot_p in the parent joint frame
 *   - <foot>_ref (src/robot-handler.cpp:29-52) is attached to the base; foot_ref_p is its
 *     placement in the base frame = base^-1 * foot at the reference configuration
 */
#ifndef SMPC_ROBOT_H
#define SMPC_ROBOT_H

#ifdef __cplusplus
extern "C" {
#endif

#define SMPC_MAX_JOINTS 32
#define SMPC_MAX_FEET 4
#define SMPC_MAX_NQ (SMPC_MAX_JOINTS + 6)
#define SMPC_NAME_LEN 32

typedef struct smpc_robot_model
{
  char name[SMPC_NAME_LEN];
  int njoints; /* including the free-flyer (joint 0) */
  int nq;      /* 7 + (njoints-1) */
  int nv;      /* 6 + (njoints-1) */
  int parent[SMPC_MAX_JOINTS]; /* -1 for joint 0 */
  int jtype[SMPC_MAX_JOINTS];  /* 0 free-flyer, 1 RX, 2 RY, 3 RZ */
  double jp_R[SMPC_MAX_JOINTS][9]; /* joint placement in parent joint frame, row-major */
  double jp_p[SMPC_MAX_JOINTS][3];
  double mass[SMPC_MAX_JOINTS];
  double com[SMPC_MAX_JOINTS][3];
  double inertia[SMPC_MAX_JOINTS][6];
  int nfeet;
  char foot_name[SMPC_MAX_FEET][SMPC_NAME_LEN];
  int foot_joint[SMPC_MAX_FEET];
  double foot_p[SMPC_MAX_FEET][3];
  double foot_ref_p[SMPC_MAX_FEET][3];
  double q_ref[SMPC_MAX_NQ];       /* named reference configuration ("standing") */
  double q_lo[SMPC_MAX_JOINTS];    /* joint limits, index = v index - 6 */
  double q_hi[SMPC_MAX_JOINTS];
  double total_mass;
} smpc_robot_model;

#ifdef __cplusplus
}
#endif
#endif
