#ifndef RMOCK_R_H_
#define RMOCK_R_H_
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include "R_ext/Random.h"
#endif
