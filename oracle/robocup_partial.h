/* robocup_partial.h — ORACLE (test infrastructure): RoboCup Partial observation (getAgentVision), see robocup_partial.c */
#ifndef ORACLE_ROBOCUP_PARTIAL_H
#define ORACLE_ROBOCUP_PARTIAL_H
#include "robocup.h"

#define RCP_CAP_BALL 32   /* 1 + 3 misclassified crosses + 10 random FP + FP balls near (9 + 10) robots */
#define RCP_CAP_ROB 20    /* 9 + 10 random FP */
#define RCP_CAP_GOAL 16   /* 4 + 10 */
#define RCP_CAP_CROSS 16  /* 3 + 1 misclassified ball + 10 */
#define RCP_CAP_FCROSS 28 /* 16 + 10 */
#define RCP_CAP_LINE 12   /* 11 */
#define RCP_OFF_BALL 0
#define RCP_OFF_ROB (RCP_OFF_BALL + RCP_CAP_BALL * 5)
#define RCP_OFF_GOAL (RCP_OFF_ROB + RCP_CAP_ROB * 7)
#define RCP_OFF_CROSS (RCP_OFF_GOAL + RCP_CAP_GOAL * 6)
#define RCP_OFF_FCROSS (RCP_OFF_CROSS + RCP_CAP_CROSS * 6)
#define RCP_OFF_LINE (RCP_OFF_FCROSS + RCP_CAP_FCROSS * 8)
#define RCP_OFF_TAIL (RCP_OFF_LINE + RCP_CAP_LINE * 5)
#define RCP_DIM (RCP_OFF_TAIL + 6 + 2 + 9) /* 6 list lengths, numLandMarks, ballsSeen, robotsSeen[9] = 793 */
#define PENALTY_LENGTH_ 60.0
#define PENALTY_WIDTH_ 110.0

int rc_partial_obs_dim(void);
/* one agent's row (RCP_DIM floats); returns 1 if a list overflowed its capacity */
int rc_agent_vision(const RoboCupEnv* e, int agentIdx, int noiseType, double magn, uint32_t tkey, float* out);
/* all agents of one snapshot: out [R][RCP_DIM]; tkey = elapsed during a step, snapshot index 0..4 at reset */
int rc_write_partial_obs(RoboCupEnv* e, uint32_t tkey, float* out);
void rcp_scene(cpv lines[11][2], double lineT[11][2], cpv crosses[3], double crossT[3][2], cpv fcross[16], double fcrossT[16][2],
               cpv posts[4], double postT[4][2]);
#endif
